"""Do a latency-bound chain of our kernels and a stream of large products overlap on two HIP streams?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, ops
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev); enc.eval(); dec.eval()
fb = synth.follower_batch(seed=0, batch=100, steps=1, n_viewpoints=64)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
A = torch.randn(2000, 2048, device=dev); Bm = torch.randn(2000, 4352, device=dev)
x = torch.randn(100, 4864, device=dev); w = torch.randn(2048, 4864, device=dev) * 0.02
side = torch.cuda.Stream()
def chain():                     # 80 dependent encoder steps (~0.75 ms)
    with torch.no_grad():
        enc(batch.seq, batch.lengths)
def big_torch():                 # torch fp32 GEMMs (~0.3 ms each)
    for _ in range(3): (A.t() @ Bm)
def big_ours():                  # our tiled gate product, 30 launches (~0.75 ms)
    for _ in range(30): ops.linear_fwd(x, w)
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def both(g):
    def f():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            g()
        chain()
        cur.wait_stream(side)
    return f
with torch.cuda.stream(side):
    big_ours(); big_torch()
torch.cuda.synchronize()
print('chain %.3f ms | torch GEMMs %.3f | ours %.3f' % (t(chain), t(big_torch), t(big_ours)))
print('chain || torch GEMMs %.3f ms   chain || ours %.3f ms' % (t(both(big_torch)), t(both(big_ours))))
hi = torch.cuda.Stream(priority=-1)
def both_hi(g):
    def f():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur); hi.wait_stream(cur)
        with torch.cuda.stream(side):
            g()
        with torch.cuda.stream(hi):
            chain()
        cur.wait_stream(side); cur.wait_stream(hi)
    return f
with torch.cuda.stream(hi):
    chain()
torch.cuda.synchronize()
print('chain on a HIGH-priority stream || torch GEMMs %.3f ms   || ours %.3f ms' % (t(both_hi(big_torch)), t(both_hi(big_ours))))
