"""Driver for a kernel timeline of the training iteration (rocprofv3 --kernel-trace -- python3 this)."""
import os, sys
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
enc, dec, _, _ = bench.build_models(101, dev)
enc.train(); dec.train()
pe = [p for p in enc.parameters() if p.requires_grad]; pd = [p for p in dec.parameters() if p.requires_grad]
flat = dp.FlatGrads(pe + pd)
oe, od = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4), optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
eng = follower.FollowerEngine(enc, dec, store)
for _ in range(int(os.environ.get('REPS', 8))):
    flat.zero(); st = eng.rollout(batch, 20, 'argmax', train=True); st.loss.backward(); oe.step(); od.step()
torch.cuda.synchronize()
