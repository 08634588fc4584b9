"""Many-row NT products (M >= 512): gemm_nt_big_kernel (LDS-tiled bf16x6) against the register-streaming gemm_nt_kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_follower_amd import ops
from speaker_follower_amd._lib import lib, kernel_profile
for M, N, K in ((8000, 512, 512), (8000, 512, 1024), (8000, 992, 512), (8000, 512, 992), (2560, 2048, 4352), (2560, 2176, 256), (991, 2048, 300)):
    x = torch.randn(M, K).cuda(); w = (torch.randn(N, K) * K ** -0.5).cuda(); b = torch.randn(N).cuda()
    res = {}
    for big in (1, 0):
        lib.sf_debug_many_row_product(big)
        for _ in range(5):
            ops.linear_fwd(x, w, b)
        torch.cuda.synchronize()
        with kernel_profile() as prof:
            for _ in range(30):
                ops.linear_fwd(x, w, b)
        torch.cuda.synchronize()
        tot = sum(v['total_us'] for v in prof.rows.values()) / 30
        res[big] = (tot, '+'.join(k.split('(')[0].replace('sf::', '')[:22] for k in prof.rows))
    lib.sf_debug_many_row_product(1)
    fl = 2.0 * M * N * K
    print('%5d x %4d x %4d  big %7.1f us (%6.1f TF)   streaming %7.1f us (%6.1f TF)   [%s | %s]' % (
        M, N, K, res[1][0], fl / res[1][0] / 1e6, res[0][0], fl / res[0][0] / 1e6, res[1][1], res[0][1]))
