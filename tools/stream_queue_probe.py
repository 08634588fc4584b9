"""Does the training iteration's two-stream backward depend on WHICH torch stream the engine gets?  HIP maps streams to a
small number of hardware queues; a side stream that shares the main stream's queue cannot overlap with it.
Runs bench.measure_train in a fresh process after creating N dummy streams first (N from argv)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch        # noqa: E402

import bench        # noqa: E402
from speaker_follower_amd import synth, features, follower    # noqa: E402


def main():
    n_dummy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    dev = torch.device('cuda', 0)
    enc, dec, _, _ = bench.build_models(101, dev)
    store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
    fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=10567)
    batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
    dummies = [torch.cuda.Stream(device=dev) for _ in range(n_dummy)]
    for s in dummies:                      # (make them real: a launch on each)
        with torch.cuda.stream(s):
            torch.zeros(4, device=dev)
    torch.cuda.synchronize()
    out = bench.measure_train(enc, dec, store, batch, 20, 20, 5)
    print('dummy streams created first: %d -> %.3f ms per training iteration' % (n_dummy, out['ms_per_iteration']))


if __name__ == '__main__':
    main()
