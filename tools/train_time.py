"""Wall time of the training iteration (student-forced rollout with dropout + BPTT + 2x Adam, B=100, 20 steps,
two-stream backward) for several weight-gradient chunkings (FollowerEngine.wgrad_chunks)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, dp, optim
dev = torch.device('cuda', 0)
NV = int(os.environ.get('NV', 10567))
store = features.FeatureStore(bench.device_table(NV, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=NV)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
for chunks in [int(c) for c in os.environ.get('CHUNKS', '1,2,4,5,10,1,4').split(',')]:
    enc, dec, _, _ = bench.build_models(101, dev)
    enc.train(); dec.train()
    pe = [p for p in enc.parameters() if p.requires_grad]; pd = [p for p in dec.parameters() if p.requires_grad]
    flat = dp.FlatGrads(pe + pd)
    oe, od = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4), optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
    eng = follower.FollowerEngine(enc, dec, store)
    eng.wgrad_chunks = chunks
    def it():
        flat.zero(); st = eng.rollout(batch, 20, 'argmax', train=True); st.loss.backward(); oe.step(); od.step(); return st
    for _ in range(5): st = it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): st = it()
    torch.cuda.synchronize()
    print('wgrad_chunks=%2d: %.3f ms per iteration, loss %.6f, |g| %.6f' % (chunks, (time.perf_counter() - t0) * 50, float(st.loss), float(flat.flat.norm())))
