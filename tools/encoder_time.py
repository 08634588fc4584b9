"""Times the follower encoder alone (B=100, T=80): persistent launch vs one launch per step, forward and
backward, with the in-process kernel timer (sf_profile_*), wall time, and the inner timeline of the
persistent kernels (sf_debug_trace: tick sums of wave 0 of every workgroup)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from speaker_follower_amd import _lib
import test_gpu_persistent as tp
enc = tp.encoder()
B = int(os.environ.get('B', 100))
seq, mask, lens = tp.batch(5, B, 10, 79)
T, H = max(lens), enc.hidden_size
fwd = tp.run(enc, seq, lens, True)
dctx = torch.randn(B, T, H, device='cuda')
z = torch.zeros(B, H, device='cuda')
for name, fn in (('forward', lambda p: tp.run(enc, seq, lens, p)),
                 ('backward', lambda p: tp.run_bwd(enc, seq, lens, fwd, p, False, dctx, z, z))):
    for persistent in (False, True):
        for _ in range(3):
            fn(persistent)
        torch.cuda.synchronize()
        with _lib.kernel_profile() as prof:
            for _ in range(5):
                fn(persistent)
        tot = sum(r['total_us'] for r in prof.rows.values()) / 5
        print('%s persistent=%s: kernel time %.1f us per call' % (name, persistent, tot))
        for k, r in sorted(prof.rows.items(), key=lambda kv: -kv[1]['total_us'])[:3]:
            print('    %-60s calls %4d avg %8.2f us (%.2f us/step)' % (k[:60], r['calls'] / 5, r['avg_us'],
                                                                       r['avg_us'] * r['calls'] / 5 / T if r['calls'] / 5 <= 2 else r['avg_us']))
    trace = torch.zeros(256 * 8, dtype=torch.int64, device='cuda')
    _lib.lib.sf_debug_trace(trace.data_ptr())
    fn(True)
    torch.cuda.synchronize()
    _lib.lib.sf_debug_trace(None)
    tr = trace.cpu().numpy().reshape(256, 8)
    t = tr[:, :5].astype(np.float64) / 100.0 / (T - 1)
    print('  %s inner timeline (us/step; %d of 256 workgroups on the one-XCD fast path)' % (name, int(tr[:, 5].sum())))
    labels = (['wait for h_t', 'MFMA + partials to LDS + barrier', 'reduce + cell + publish', '-', 'tapes + loop'] if name == 'forward'
              else ['wait for partials', 'cell bwd + tile + barrier + reset', 'MFMA + drain', 'publish issue', 'loop + fetch'])
    for k, n in enumerate(labels):
        print('    %-36s mean %.2f  min %.2f  max %.2f' % (n, t[:, k].mean(), t[:, k].min(), t[:, k].max()))
