"""Does the speaker batch time depend on what ran before in the process? (full bench: 1.77 ms, alone: 1.45 ms)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, bench_extras
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev); enc.eval(); dec.eval()
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
def spk(tag):
    out = bench_extras.speaker_decode(store, dev)
    print('%-40s greedy %.3f ms  teacher %.3f ms' % (tag, out['greedy_decode']['ms_per_batch'], out['teacher_scoring']['ms_per_batch']))
spk('fresh process')
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=10567)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
eng = follower.FollowerEngine(enc, dec, store)
replay, st = eng.capture(batch, 20, 'argmax')
for _ in range(25): replay()
torch.cuda.synchronize()
spk('after 25 follower graph replays')
streams = [torch.cuda.Stream() for _ in range(2)]
reps = []
for i, s in enumerate(streams):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        reps.append(follower.FollowerEngine(enc, dec, store).capture(batch, 20, 'argmax'))
torch.cuda.synchronize()
for k in range(20):
    with torch.cuda.stream(streams[k % 2]):
        reps[k % 2][0]()
torch.cuda.synchronize()
spk('after two-stream in-flight rollouts')
from speaker_follower_amd import _lib
with torch.no_grad():
    with _lib.kernel_profile() as prof:
        for _ in range(3): eng.rollout(batch, 20, 'argmax', train=False)
spk('after profiled eager rollouts')
