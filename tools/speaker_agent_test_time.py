"""Seq2SeqSpeaker.test (speaker.py:397-414; what data_augmentation_from_speaker.py drives: greedy instructions for the paths of
an environment) through the agents' API on the full world, minibatches of 100: paths per second, milliseconds per minibatch
(the device's greedy decode of one minibatch alone: ~1.3 ms), with a cProfile of the host side (--cprofile)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
CPROFILE = '--cprofile' in sys.argv
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import bench_extras, features, agents
dev = torch.device('cuda', 0)
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
senc, sdec = bench_extras._speaker_models(dev)
e, _ = bench_extras.full_world(store, 100, seed=33, n_items=2000)
spk = agents.Seq2SeqSpeaker(e, '/tmp/spk_test.json', senc, sdec, 80)
spk.store = store
for after, name in ((10 ** 9, 'loop over rollout()'), (0, 'one sweep (first call: captures its graphs)'), (0, 'one sweep'), (0, 'one sweep')):
    spk.sweep_test_after = after
    if after:
        spk.test(use_dropout=False, feedback='argmax')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = spk.test(use_dropout=False, feedback='argmax')
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('speaker.test over %d paths, %-44s %.3f s = %6.0f paths/s, %.2f ms per minibatch of 100'
          % (len(res), name + ':', dt, len(res) / dt, dt / (len(res) / 100) * 1e3))
if CPROFILE:
    pr = cProfile.Profile(); pr.enable(); spk.test(use_dropout=False, feedback='argmax'); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
