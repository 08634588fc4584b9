"""Do independent branches of ONE hipGraph run concurrently on this stack?  Two chains of small
dependent kernels on two captured streams: serial ~ 2x one chain, parallel ~ 1x."""
import os, time, torch
dev = torch.device('cuda', 0)
a = torch.randn(256, 256, device=dev); b = torch.randn(256, 256, device=dev)
def chain(x, n=40):
    for _ in range(n):
        x = torch.tanh(x @ x * 1e-3)
    return x
def timeit(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
side = torch.cuda.Stream()
chain(a); chain(b)
with torch.cuda.stream(side):
    chain(b)
torch.cuda.synchronize()
# one chain
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    y = chain(a)
# two chains, one stream
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    y1 = chain(a); y2 = chain(b)
# two chains, two captured streams (fork / join)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        z2 = chain(b)
    z1 = chain(a)
    cur.wait_stream(side)
print('one chain        %.1f us' % timeit(g1.replay))
print('two, one stream  %.1f us' % timeit(g2.replay))
print('two, two streams %.1f us (graph branches)' % timeit(g3.replay))
def eager2():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain(b)
    chain(a)
    cur.wait_stream(side)
print('two, two streams %.1f us (eager)' % timeit(eager2))
print('two, one stream  %.1f us (eager)' % timeit(lambda: (chain(a), chain(b))))
