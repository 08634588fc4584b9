"""Do the two chains of the backward through time -- heads (scoring / text-attention backward) and tails (LSTM / visual
attention backward) -- overlap?  Wall-clock answer WITHOUT a profiler (rocprofv3 --kernel-trace serialises the queues:
its timelines show every head before the first tail): the backward through time alone as one graph (both chains, the
product's form), the heads alone, the tails alone (sf_debug_bptt_part), and the two as separately captured graphs on two
streams.  Round 5, MI355X: 1.46 / 0.93 / 1.14 / 1.43 ms -- they overlap (the sum would be 2.07), equally well inside one
graph; max(heads, tails) = 1.14 is the floor.  Timing only: the tails-only graph reads whatever the last complete
backward left in the head outputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, _lib, runtime
dev = torch.device('cuda', 0)
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=10567)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
enc, dec, _, _ = bench.build_models(101, dev)
enc.train(); dec.train()
eng = follower.FollowerEngine(enc, dec, store)
eng._bptt_only = True            # _backward returns behind the backward through time
st = eng.rollout(batch, 20, 'argmax', train=True)
one = torch.ones((), device=dev)
with torch.no_grad():
    eng._backward(st, one)                             # both chains once, eagerly (streams, workspaces, caches)
torch.cuda.synchronize()

def capture(part, stream):
    _lib.lib.sf_debug_bptt_part(part)
    runtime.ensure_workspace(stream, dev)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(stream):
            with torch.no_grad():
                eng._backward(st, one)                      # warm-up on this stream
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=stream):
                    eng._backward(st, one)
    finally:
        _lib.lib.sf_debug_bptt_part(0)
    torch.cuda.synchronize()
    return g

for name in ('fused_cell_backward', 'slab_consumers', 'bptt_flags'):
    if 'SF_' + name.upper() in os.environ:
        getattr(_lib.lib, 'sf_debug_' + name)(int(os.environ['SF_' + name.upper()]))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g_both = capture(0, s1)
g_heads = capture(1, s1)
g_tails = capture(2, s2)

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def both_streams():
    with torch.cuda.stream(s1): g_heads.replay()
    with torch.cuda.stream(s2): g_tails.replay()
def one_graph():
    with torch.cuda.stream(s1): g_both.replay()
def heads():
    with torch.cuda.stream(s1): g_heads.replay()
def tails():
    with torch.cuda.stream(s2): g_tails.replay()
print('one graph, both chains (today):      %.3f ms' % timed(one_graph))
print('heads alone:                         %.3f ms' % timed(heads))
print('tails alone:                         %.3f ms' % timed(tails))
print('two graphs on two streams, together: %.3f ms' % timed(both_streams))
