"""Device time of the speaker's teacher-forced scoring of ONE big batch (the pragmatic re-ranking scores all ~2 500
candidate routes of a minibatch at once): per-step kernels against chunks of 128 through the persistent word loop."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import torch                # noqa: E402

import bench                # noqa: E402
from speaker_follower_amd import bench_extras, features, synth, speaker    # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    table = bench.device_table(10567, 1234, dev)
    store = features.FeatureStore(table, device=dev)
    senc, sdec = bench_extras._speaker_models(dev)
    eng = speaker.SpeakerEngine(senc, sdec, store)
    for B in (128, 512, 2560):
        sb = synth.speaker_batch(seed=3, batch=B, n_viewpoints=10567, min_path=4, max_path=7, min_len=10, max_len=79)
        b = speaker.DeviceSpeakerBatch.from_synth(sb, device=dev)
        for S in (80,):
            with torch.no_grad():
                for _ in range(2):
                    st = eng.score(b, S, 'teacher', train=False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    st = eng.score(b, S, 'teacher', train=False)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
            print('B = %4d, S = %d, persistent = %s: %.2f ms per batch (%.1f us per candidate)'
                  % (B, S, st.persistent, 1e3 * dt, 1e6 * dt / B))


if __name__ == '__main__':
    main()
