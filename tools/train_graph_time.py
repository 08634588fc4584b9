"""Training iterations eager vs replayed (follower configs[1] shape, speaker 100 x 80)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, bench_extras
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=10567)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
t = bench.measure_train(enc, dec, store, batch, 20, 10, 5)
print('follower: eager %.3f ms, graph replay %.3f ms (loss %.4f)' % (t['eager']['ms_per_iteration'], t['ms_per_iteration'], t['loss']))
s = bench_extras.speaker_train_iteration(store, dev)
print('speaker: eager %.3f ms (host issue %.3f), graph %.3f ms (host %.3f)' % (s['eager']['ms_per_iteration'], s['eager']['ms_host_issue'], s['ms_per_iteration'], s['ms_host_issue']))
