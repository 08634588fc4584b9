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
store = features.FeatureStore(bench.device_table(512, 1234, device), device=device)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=512)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=device, row0=0)
eng = follower.FollowerEngine(enc, dec, store)
trace = torch.zeros(256 * 16 * 8, dtype=torch.int64, device=device)
lib = _lib.lib
lib.sf_debug_trace_lstm.argtypes = [ctypes.c_void_p]
lib.sf_debug_trace_lstm.restype = None
with torch.no_grad():
    for _ in range(3):
        eng.rollout(batch, 20, 'argmax', train=False)
    lib.sf_debug_trace_lstm(trace.data_ptr())
    eng.rollout(batch, 20, 'argmax', train=False)
    torch.cuda.synchronize()
    lib.sf_debug_trace_lstm(None)
t = trace.cpu().numpy().reshape(256 * 16, 8).astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = (t[:, :5] - t0) / 100.0
us[us < 0] = np.nan
print('%d waves (last encoder step); us after the first block start (mean / min / max)' % len(t))
for k, n in enumerate(['start', 'loads landed', 'mfma done', 'gates summed', 'end']):
    print('%-14s %6.2f %6.2f %6.2f' % (n, np.nanmean(us[:, k]), np.nanmin(us[:, k]), np.nanmax(us[:, k])))
