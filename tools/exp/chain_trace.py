"""Development aid: timeline of the chained launch (visual partials || h~ -> t_a)."""
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
nblk = 1024
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
live = t[:, 0] > 0
t0 = t[live, 0].min()
role = t[:, 7]
def show(name, m, cols):
    if not m.any():
        return
    us = (t[m][:, cols] - t0) / 100.0
    print('%-10s n=%3d ' % (name, m.sum()) + '  '.join('%s %5.1f/%5.1f/%5.1f' % (c, us[:, i].mean(), us[:, i].min(), us[:, i].max())
                                                       for i, c in enumerate(['start', 'wait', 'mfma', 'end'][:len(cols)] if name != 'vis' else ['start', 'rows', 'stored'])))
show('vis', live & (role == 0), [0, 1, 2])
show('producer', live & (role == 1), [0, 2, 3])
show('consumer', live & (role == 2), [0, 1, 2, 3])
