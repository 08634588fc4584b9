"""Where a training iteration spends its time (synchronised phase timers; sum > the pipelined
iteration because the syncs remove overlap across phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, dp, optim
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
enc.train(); dec.train()
store = features.FeatureStore(bench.device_table(2048, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=2048)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
pe = [p for p in enc.parameters() if p.requires_grad]; pd = [p for p in dec.parameters() if p.requires_grad]
flat = dp.FlatGrads(pe + pd)
oe, od = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4), optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
eng = follower.FollowerEngine(enc, dec, store)
eng.two_stream_backward = os.environ.get('TWO', '1') == '1'
def sync(): torch.cuda.synchronize(); return time.perf_counter()
acc = {}
for it in range(13):
    t0 = sync(); flat.zero(); t1 = sync()
    st = eng.rollout(batch, 20, 'argmax', train=True); t2 = sync()
    st.loss.backward(); th = time.perf_counter(); t3 = sync()
    if it >= 3: acc['backward_host_enqueue'] = acc.get('backward_host_enqueue', 0) + (th - t2) / 10
    oe.step(); od.step(); t4 = sync()
    if it >= 3:
        for k, v in (('zero', t1 - t0), ('forward', t2 - t1), ('backward', t3 - t2), ('adam', t4 - t3)):
            acc[k] = acc.get(k, 0) + v / 10
print({k: round(v * 1e3, 3) for k, v in acc.items()}, 'ms')
