"""Per-kernel table of one speaker batch (100 paths x 80 words, greedy), measured with the in-process timer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import features, bench_extras
dev = torch.device('cuda', 0)
store = features.FeatureStore(bench.device_table(int(os.environ.get('NV', 10567)), 1234, dev), device=dev)
out = bench_extras.speaker_decode(store, dev)
print(out['greedy_decode'], out['roofline'])
for r in out['kernels']:
    print('%-44s calls %5.1f avg %8.2f us  per batch %8.1f us  %5.1f%%' % (r['kernel'][:44], r['calls_per_run'], r['avg_us'], r['us_per_run'], 100 * r['share']))
