"""Per-kernel time of one training iteration (student-forced rollout with dropout + BPTT + 2x Adam,
B=100, 20 steps) with the in-process kernel timer, single stream (so kernel times add up)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, dp, optim, _lib
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
enc.train(); dec.train()
NV = int(os.environ.get('NV', 10567))
store = features.FeatureStore(bench.device_table(NV, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=NV)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
pe = [p for p in enc.parameters() if p.requires_grad]; pd = [p for p in dec.parameters() if p.requires_grad]
flat = dp.FlatGrads(pe + pd)
oe, od = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4), optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
eng = follower.FollowerEngine(enc, dec, store)
eng.two_stream_backward = False
def it():
    flat.zero(); st = eng.rollout(batch, 20, 'argmax', train=True); st.loss.backward(); oe.step(); od.step()
for _ in range(5): it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): it()
torch.cuda.synchronize()
print('one-stream iteration %.3f ms' % ((time.perf_counter() - t0) * 100))
with _lib.kernel_profile() as prof:
    for _ in range(3): it()
tot = sum(r['total_us'] for r in prof.rows.values()) / 3
print('kernel time per iteration %.1f us' % tot)
for k, r in sorted(prof.rows.items(), key=lambda kv: -kv[1]['total_us'])[:28]:
    print('%-64s calls %6.1f avg %8.2f us  total %8.1f us  %5.1f%%' % (k[:64], r['calls'] / 3, r['avg_us'], r['total_us'] / 3, 100 * r['total_us'] / 3 / tot))
