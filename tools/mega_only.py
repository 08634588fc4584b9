"""Runs the persistent decode rollout REPS times (for rocprofv3 --pmc / --kernel-trace passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import test_gpu_mega as T
from speaker_follower_amd import synth, follower
B, S = int(os.environ.get('B', 100)), int(os.environ.get('S', 20))
eng = T.make_engine()
batch = follower.DeviceFollowerBatch.from_synth(synth.follower_batch(seed=47, batch=B, steps=S, n_viewpoints=256))
eng.persistent_decode = os.environ.get('PERSISTENT', '1') == '1'
with torch.no_grad():
    for _ in range(int(os.environ.get('REPS', 4))):
        st = eng.rollout(batch, S, 'argmax', train=False)
    torch.cuda.synchronize()
print('persistent', getattr(st, 'persistent', None))
