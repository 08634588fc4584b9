"""BASELINE configs[4] through the agents' own API on the FULL world, per minibatch of 64 instructions
(rational_follower.py:35-148): state_factored_search(K = 40, 1) -> every candidate route scored by the speaker with
teacher forcing (Seq2SeqSpeaker._score_obs_actions_and_instructions over ALL candidates at once, as the reference does)
-> rational_mix.  Every minibatch is one the process has not seen before.

    python tools/pragmatic_profile.py [--cprofile]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import torch                # noqa: E402

import bench                # noqa: E402
from speaker_follower_amd import bench_extras, features    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cprofile', action='store_true')
    ap.add_argument('--minibatches', type=int, default=6)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    enc, dec, _, _ = bench.build_models(101, dev)
    table = bench.device_table(10567, 1234, dev)
    store = features.FeatureStore(table, device=dev)
    pr = cProfile.Profile() if args.cprofile else None
    out = bench_extras.pragmatic_inference(enc, dec, store, dev, minibatches=args.minibatches, profiler=pr)
    for k, v in out.items():
        print('%-28s %s' % (k, v))
    if pr is not None:
        pstats.Stats(pr).sort_stats('tottime').print_stats(30)


if __name__ == '__main__':
    main()
