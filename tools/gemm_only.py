"""Runs only the decoder-LSTM-gate product (the bench's roofline kernel) a few times: target
for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_follower_amd import ops
B = int(os.environ.get('B', 100))
x = torch.randn(B, 4864, device='cuda'); w = torch.randn(2048, 4864, device='cuda') * 0.02
for _ in range(int(os.environ.get('REPS', 20))):
    y = ops.linear_fwd(x, w)
torch.cuda.synchronize()
print('ok', float(y.abs().mean()))
