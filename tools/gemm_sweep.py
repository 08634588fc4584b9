"""Times the gate-product kernel path (sf_linear_slabs_fwd) over batch sizes: how the stage time
moves with the A-tile height tells MFMA-bound from load-bound.  Run on the GPU box."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_follower_amd import _lib, runtime
lib = _lib.lib
K1, K2, N = 4352, 512, 2048
w = torch.randn(N, K1, device='cuda') * 0.02
u = torch.randn(N, K2, device='cuda') * 0.02
ws = torch.zeros(lib.sf_workspace_bytes(), dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for M in [int(v) for v in os.environ.get('MS', '16,32,48,64,80,96,100,112,128').split(',')]:
    x = torch.randn(M, K1, device='cuda'); h = torch.randn(M, K2, device='cuda')
    ks = C.c_int(0)
    def run():
        _lib.call('sf_linear_slabs_fwd', x.data_ptr(), K1, w.data_ptr(), K1, h.data_ptr(), K2, u.data_ptr(), K2,
                  M, N, C.byref(ks), ws.data_ptr(), ws.numel(), st)
    for _ in range(10): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print('M=%3d ks=%d  %.2f us/launch  %.1f TFLOP/s' % (M, ks.value, us, 2.0 * M * N * (K1 + K2) / us * 1e-6))
