"""Lab: do the decode step's small products care about the row stride of their activation operand?  (sf_linear_fwd takes
ldx / ldy: the same kernels with x [100, K] contiguous (power-of-two strides: 1 KB, 2 KB, 4 KB) and padded by 16 floats.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speaker_follower_amd._lib import call, kernel_profile
from speaker_follower_amd.runtime import ptr, ws_args
M = 100
junk = torch.empty(32 << 20, device='cuda')
for K, N, what in ((256, 2176, 'r = W_a^T wt'), (512, 256, 't_a / t_v'), (512, 512, 't_text'), (1024, 512, 'h~'), (512, 2180, 'folded r / q'),
                   (256, 2176, 'q = W_v^T t_v')):
    w = (torch.randn(N, K) * K ** -0.5).cuda(); b = torch.randn(N).cuda()
    line = '%-16s [100 x %4d] x [%4d x %4d]^T:' % (what, K, N, K)
    for pad_x, pad_y in ((0, 0), (16, 0), (0, 16), (16, 16), (8, 8), (32, 32)):
        x = torch.randn(M, K + pad_x).cuda(); y = torch.empty(M, N + pad_y, device='cuda')
        for _ in range(5):
            call('sf_linear_fwd', ptr(x), K + pad_x, ptr(w), ptr(b), M, N, K, 0, ptr(y), N + pad_y, *ws_args(x.device))
        torch.cuda.synchronize()
        with kernel_profile() as prof:
            for _ in range(40):
                junk.zero_()
                call('sf_linear_fwd', ptr(x), K + pad_x, ptr(w), ptr(b), M, N, K, 0, ptr(y), N + pad_y, *ws_args(x.device))
        torch.cuda.synchronize()
        rows = {k: v for k, v in prof.rows.items() if 'gemm' in k}
        us = sum(v['total_us'] for v in rows.values()) / 40
        line += '  x+%d y+%d %.2f' % (pad_x, pad_y, us)
    print(line + '   [%s]' % ' + '.join(k.split('(')[0].replace('sf::', '') for k in rows))
