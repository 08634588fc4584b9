"""Times the follower encoder alone (B=100, T=80): persistent launch vs one launch per step, with the
in-process kernel timer (sf_profile_*) and with wall time over back-to-back calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from speaker_follower_amd import _lib
import test_gpu_persistent as tp
enc = tp.encoder()
B = int(os.environ.get('B', 100))
seq, mask, lens = tp.batch(5, B, 10, 79)
for persistent in (False, True):
    for _ in range(5):
        tp.run(enc, seq, lens, persistent)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        tp.run(enc, seq, lens, persistent)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50
    with _lib.kernel_profile() as prof:
        for _ in range(10):
            tp.run(enc, seq, lens, persistent)
    tot = sum(r['total_us'] for r in prof.rows.values()) / 10
    print('persistent=%s: wall %.1f us per encoder call, kernel time %.1f us (T=%d -> %.2f us/step)'
          % (persistent, wall * 1e6, tot, max(lens), tot / max(lens)))
    for k, r in sorted(prof.rows.items(), key=lambda kv: -kv[1]['total_us'])[:4]:
        print('    %-60s calls %4d avg %8.2f us' % (k[:60], r['calls'] / 10, r['avg_us']))
# inner timeline of the persistent kernel (sf_debug_trace: tick sums of wave 0 of every workgroup)
import numpy as np
trace = torch.zeros(256 * 8, dtype=torch.int64, device='cuda')
_lib.lib.sf_debug_trace(trace.data_ptr())
tp.run(enc, seq, lens, True)
torch.cuda.synchronize()
_lib.lib.sf_debug_trace(None)
t = trace.cpu().numpy().reshape(256, 8)[:, :5].astype(np.float64) / 100.0 / (max(lens) - 1)   # us per step
for k, n in enumerate(['wait for h_t', 'MFMA + partials to LDS + barrier', 'reduce + cell + publish', '(unused)', 'tapes + loop']):
    print('%-36s mean %.2f  min %.2f  max %.2f us/step' % (n, t[:, k].mean(), t[:, k].min(), t[:, k].max()))
