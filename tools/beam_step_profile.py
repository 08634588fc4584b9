"""Kernel-level view of ONE flat decoder step over 64 x 40 = 2 560 search states (bench_extras.search_step: beam_64x40).
    rocprofv3 --kernel-trace --stats -- python3 tools/beam_step_profile.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                # noqa: E402

import bench                # noqa: E402
from speaker_follower_amd import bench_extras, features    # noqa: E402

dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
enc.eval()
dec.eval()
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
out = bench_extras.search_step(enc, dec, store, dev)
print({k: v.get('ms_per_step', v.get('ms_total')) for k, v in out.items() if isinstance(v, dict)})
