"""speaker.SpeakerSweep (BASELINE configs[2]: greedy decoding of many path minibatches) with ONE, two and three streams:
milliseconds per minibatch of 100 over 400 minibatches, words identical.  VERDICT round 5: in the whole-bench profile the
second stream's first kernel (gather_path_actions_kernel) averages 704 us -- does the second stream overlap anything?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import bench_extras, features, speaker, synth
dev = torch.device('cuda', 0)
N = int(os.environ.get('SF_SWEEP_BATCHES', '400'))
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
enc, dec = bench_extras._speaker_models(dev)
sbs = [synth.speaker_batch(seed=500 + i, batch=100, n_viewpoints=10567, min_path=4, max_path=7, min_len=10, max_len=79)
       for i in range(N)]
ref = None
STREAMS = tuple(int(x) for x in os.environ.get('SF_SWEEP_STREAMS', '1,2,3').split(','))
for rep in range(2):
    for ns in STREAMS:
        sweep = speaker.SpeakerSweep(enc, dec, store, 100, 80, n_streams=ns)
        sweep.run(sbs[:40])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = sweep.run(sbs)
        dt = time.perf_counter() - t0
        if ref is None:
            ref = out.copy()
        print('streams %d: %.3f ms per minibatch, %6.0f paths/s, host packing %.3f s of %.3f s, words equal: %s, fallbacks %d'
              % (ns, 1e3 * dt / N, 100 * N / dt, sweep.host_pack_s, dt, bool((out == ref).all()), sweep.fallbacks), flush=True)
