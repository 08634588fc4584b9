"""Driver for a kernel timeline of the two-stream forward (rocprofv3 --kernel-trace -- python3 this):
a few eager inference rollouts, nothing else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
enc.eval(); dec.eval()
NV = int(os.environ.get('NV', 10567))
store = features.FeatureStore(bench.device_table(NV, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=NV)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
eng = follower.FollowerEngine(enc, dec, store)
eng.two_stream_forward = os.environ.get('TWO', '1') == '1'
if os.environ.get('GRAPH', '1') == '1':
    replay, gst = eng.capture(batch, 20, 'argmax')
    for _ in range(int(os.environ.get('REPS', 6))):
        replay()
    torch.cuda.synchronize()
else:
    with torch.no_grad():
        for _ in range(int(os.environ.get('REPS', 6))):
            eng.rollout(batch, 20, 'argmax', train=False)
        torch.cuda.synchronize()
