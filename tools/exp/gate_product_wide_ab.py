"""(runs with tools/exp/gate_product_wide.patch applied)  A/B (round 5): the decoder's gate product with 64-column blocks x 8 K-splits (gemm_nt_split_kernel) vs 128-column blocks x
16 K-splits (gemm_nt_split_wide_kernel): correctness against float64, kernel time, cell-kernel time, whole rollout."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np                                             # noqa: E402
import torch                                                   # noqa: E402
from speaker_follower_amd._lib import call, lib, kernel_profile       # noqa: E402
from speaker_follower_amd.runtime import ptr, ws_args, workspace      # noqa: E402

N, K1, K2 = 2048, 4352, 512
dev = torch.device('cuda', 0)
for M in (100, 37, 128):
    g = torch.Generator().manual_seed(M)
    x, h = torch.randn(M, K1, generator=g).cuda(), torch.randn(M, K2, generator=g).cuda()
    w, u = (torch.randn(N, K1, generator=g) * 0.02).cuda(), (torch.randn(N, K2, generator=g) * 0.02).cuda()
    ref = x.double().cpu() @ w.double().cpu().T + h.double().cpu() @ u.double().cpu().T
    for wide in (0, 1):
        lib.sf_debug_gate_product_wide(wide)
        ks = C.c_int(0)
        for _ in range(20):
            call('sf_linear_slabs_fwd', ptr(x), K1, ptr(w), K1, ptr(h), K2, ptr(u), K2, M, N, C.byref(ks), *ws_args(dev))
        torch.cuda.synchronize()
        slabs = workspace(dev).view(torch.float32)[:ks.value * M * N].view(ks.value, M, N)
        got = slabs.double().sum(0).cpu()
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        with kernel_profile() as prof:
            for _ in range(200):
                call('sf_linear_slabs_fwd', ptr(x), K1, ptr(w), K1, ptr(h), K2, ptr(u), K2, M, N, C.byref(ks), *ws_args(dev))
        torch.cuda.synchronize()
        (name, r), = [(k, v) for k, v in prof.rows.items() if 'gemm_nt' in k]
        print('M=%3d wide=%d  %-34s ks=%2d  avg %6.2f us  min %6.2f us   max error / scale %.2e' % (M, wide, name, ks.value, r['avg_us'], r['min_us'], err))
lib.sf_debug_gate_product_wide(0)

# whole rollout
sys.argv = ['bench.py']
import bench                                                   # noqa: E402
from speaker_follower_amd import synth, features, follower     # noqa: E402
enc, dec, _, _ = bench.build_models(101, dev)
enc.eval(); dec.eval()
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=10567)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
import time
acts = {}
for wide in (0, 1, 0, 1):
    lib.sf_debug_gate_product_wide(wide)
    eng = follower.FollowerEngine(enc, dec, store)
    replay, st = eng.capture(batch, 20, 'argmax')
    for _ in range(10):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    acts[wide] = st.actions.cpu().numpy().copy()
    with torch.no_grad(), kernel_profile() as prof:
        eng.rollout(batch, 20, 'argmax', train=False)
    torch.cuda.synchronize()
    rows = {k: v for k, v in prof.rows.items() if 'gemm_nt_split' in k or 'lstm_pw_fwd' in k}
    print('wide=%d  rollout %.4f ms (%.0f agent-steps/s)  %s' % (wide, 1e3 * dt, 2000 / dt, {k[:32]: round(v['avg_us'], 2) for k, v in rows.items()}))
lib.sf_debug_gate_product_wide(0)
print('actions equal:', bool(np.array_equal(acts[0], acts[1])))
