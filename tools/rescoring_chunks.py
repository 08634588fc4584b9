"""Speaker rescoring of 2560 candidate paths (configs[4]: 64 instructions x K = 40), teacher-forced, 80 words:
time vs minibatch size (128 = the persistent word-loop kernel's limit; larger = the per-step kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, speaker, bench_extras
dev = torch.device('cuda', 0)
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
senc, sdec = bench_extras._speaker_models(dev)
eng = speaker.SpeakerEngine(senc, sdec, store)
n = 2560
for chunk in (128, 256, 640, 1280, 2560):
    sbs = [synth.speaker_batch(seed=100 + i, batch=chunk, n_viewpoints=10567, min_path=4, max_path=7, min_len=10, max_len=79)
           for i in range(n // chunk)]
    def rescore():
        with torch.no_grad():
            return torch.cat([eng.score(speaker.DeviceSpeakerBatch.from_synth(sb, device=dev), 80, 'teacher').step_scores.sum(0)
                              for sb in sbs])
    rescore(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): rescore()
    torch.cuda.synchronize()
    print('chunk %4d: %.2f ms per 2560 candidates' % (chunk, (time.perf_counter() - t0) / 3 * 1e3))
