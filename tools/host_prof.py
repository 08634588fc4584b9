import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from speaker_follower_amd import synth, model, features, follower
d = synth.FULL
enc_w, dec_w = synth.follower_weights(0)
enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()}); dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
enc.cuda().eval(); dec.cuda().eval()
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=512)
store = features.FeatureStore(synth.feature_table(0, 512))
batch = follower.DeviceFollowerBatch.from_synth(fb)
eng = follower.FollowerEngine(enc, dec, store)
with torch.no_grad():
    for _ in range(3): eng.rollout(batch, 20, 'argmax', train=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): eng.rollout(batch, 20, 'argmax', train=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('enqueue %.3f ms per rollout, +sync %.3f ms total per rollout' % ((t1 - t0) * 100, (t2 - t0) * 100))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): eng.rollout(batch, 20, 'argmax', train=False)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
    for n in (5, 20, 40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): st = eng.rollout(batch, 20, 'argmax', train=False)
        torch.cuda.synchronize()
        print('%d rollouts: %.3f ms each' % (n, (time.perf_counter() - t0) * 1e3 / n))
