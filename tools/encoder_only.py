"""Runs the follower encoder (80 recurrent steps, batch 100) plus ONE decode step a few times:
target for rocprofv3 --pmc passes over lstm_step_wide_kernel (filter by kernel name)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower
device = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, device)
enc.eval(); dec.eval()
store = features.FeatureStore(bench.device_table(64, 1234, device), device=device)
fb = synth.follower_batch(seed=0, batch=100, steps=1, n_viewpoints=64)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=device, row0=0)
eng = follower.FollowerEngine(enc, dec, store)
with torch.no_grad():
    for _ in range(int(os.environ.get('REPS', 5))):
        eng.rollout(batch, 1, 'argmax', train=False)
torch.cuda.synchronize()
print('ok')
