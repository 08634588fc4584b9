"""Development aid: per-block timeline of the visual-attention partials body inside the paired launch.
Run on the GPU box: python tools/vis_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, _lib

device = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, device)
enc.eval(); dec.eval()
NV = int(os.environ.get('NV', '10567'))
store = features.FeatureStore(bench.device_table(NV, 1234, device), device=device)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=NV)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=device, row0=0)
eng = follower.FollowerEngine(enc, dec, store)
nblk = 512
trace = torch.zeros(nblk * 8, dtype=torch.int64, device=device)
lib = _lib.lib
with torch.no_grad():
    for _ in range(3):
        eng.rollout(batch, 20, 'argmax', train=False)
    lib.sf_debug_trace(trace.data_ptr())
    eng.rollout(batch, 20, 'argmax', train=False)
    torch.cuda.synchronize()
    lib.sf_debug_trace(None)
t = trace.cpu().numpy().reshape(nblk, 8).astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = (t[:, :3] - t0) / 100.0          # wall_clock64: 100 MHz
print('%d blocks; us after the first block start (mean / min / max)' % len(t))
for k, n in enumerate(['start', 'rows loaded + scores', 'partials stored']):
    print('%-22s %6.1f %6.1f %6.1f' % (n, us[:, k].mean(), us[:, k].min(), us[:, k].max()))
