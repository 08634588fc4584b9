import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_follower_amd import _lib
from speaker_follower_amd.runtime import ptr, stream
print({k: v for k, v in os.environ.items() if any(s in k for s in ('HIP', 'HSA', 'AMD', 'ROC', 'GPU'))})
x = torch.zeros(1 << 20, device='cuda')
for n in (64, 51200, 1 << 20):
    for _ in range(10):
        _lib.call('sf_fill_f32', ptr(x), n, 1.0, stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 2000
    e0.record()
    for _ in range(R):
        _lib.call('sf_fill_f32', ptr(x), n, 1.0, stream())
    e1.record()
    torch.cuda.synchronize()
    print('eager fill n=%d: %.2f us/kernel' % (n, e0.elapsed_time(e1) * 1e3 / R))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(200):
                _lib.call('sf_fill_f32', ptr(x), n, 1.0, C_stream := stream())
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print('graph fill n=%d: %.2f us/kernel' % (n, e0.elapsed_time(e1) * 1e3 / 2000))
y = torch.zeros(1 << 20, device='cuda')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(2000):
    y.add_(1.0)
e1.record(); torch.cuda.synchronize()
print('torch add_ 1M: %.2f us/kernel' % (e0.elapsed_time(e1) * 1e3 / 2000))
