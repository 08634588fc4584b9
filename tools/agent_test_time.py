"""Seq2SeqAgent.test (follower.py:987-999: argmax rollouts over a whole split, results dictionary) through the agents' API
on the full world, minibatches of 100: instructions per second and milliseconds per minibatch, with a cProfile of the
host side."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
CPROFILE = '--cprofile' in sys.argv
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import bench_extras, features, agents
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
e, nt = bench_extras.full_world(store, 100, seed=41, n_items=2000)
ag = agents.Seq2SeqAgent(e, '/tmp/agent_test.json', enc, dec, episode_len=20)
ag.store = store
ag.use_device_env(nt)
ag.test(use_dropout=False, feedback='argmax')              # first epoch: hop tables, caches
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = ag.test(use_dropout=False, feedback='argmax')
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('agent.test over %d instructions: %.3f s = %.0f instructions/s, %.2f ms per minibatch of 100 (device rollout alone: ~2.0 ms)'
          % (len(res), dt, len(res) / dt, dt / (len(res) / 100) * 1e3))
if CPROFILE:
    pr = cProfile.Profile(); pr.enable(); ag.test(use_dropout=False, feedback='argmax'); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
