"""A/B of the decoder's gate product [B,4864] x [2048,4864]^T: fp32 MFMA (rounds 1-3) vs bf16 x 6 splitting.
    python tools/gate_product_ab.py [B ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                   # noqa: E402
from speaker_follower_amd._lib import call, lib, kernel_profile       # noqa: E402
from speaker_follower_amd.runtime import ptr, ws_args          # noqa: E402

N, K1, K2 = 2048, 4352, 512
for M in [int(a) for a in sys.argv[1:]] or [100, 16, 32, 64, 128]:
    x, h = torch.randn(M, K1).cuda(), torch.randn(M, K2).cuda()
    w, u = (torch.randn(N, K1) * 0.02).cuda(), (torch.randn(N, K2) * 0.02).cuda()
    res = {}
    for f32 in (1, 0):
        lib.sf_debug_gate_product_f32(f32)
        ks = C.c_int(0)
        for _ in range(20):
            call('sf_linear_slabs_fwd', ptr(x), K1, ptr(w), K1, ptr(h), K2, ptr(u), K2, M, N, C.byref(ks), *ws_args(x.device))
        torch.cuda.synchronize()
        with kernel_profile() as prof:
            for _ in range(200):
                call('sf_linear_slabs_fwd', ptr(x), K1, ptr(w), K1, ptr(h), K2, ptr(u), K2, M, N, C.byref(ks), *ws_args(x.device))
        torch.cuda.synchronize()
        (name, r), = [(k, v) for k, v in prof.rows.items() if 'gemm_nt' in k]
        res[f32] = (name, r['avg_us'], r['min_us'])
    lib.sf_debug_gate_product_f32(0)
    fl = 2.0 * M * N * (K1 + K2)
    print('M=%3d  %-28s avg %6.2f us min %6.2f (%.1f TF)   %-28s avg %6.2f us min %6.2f (%.1f TF algorithmic)' % (
        M, res[1][0], res[1][1], res[1][2], fl / res[1][1] / 1e6, res[0][0], res[0][1], res[0][2], fl / res[0][1] / 1e6))
