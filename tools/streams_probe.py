"""Throughput of N independent rollout graphs replayed concurrently on N streams."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_follower_amd import synth, model, features, follower
from speaker_follower_amd import runtime
d = synth.FULL
dev = torch.device('cuda')
enc_w, dec_w = synth.follower_weights(101)
enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight']).cuda().eval()
dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat).cuda().eval()
table = torch.rand(2048, 36, 2048, device=dev)
store = features.FeatureStore(table)
for NS in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(NS)]
    reps = []
    for i, s in enumerate(streams):
        fb = synth.follower_batch(seed=i, batch=100, steps=20, n_viewpoints=2048)
        batch = follower.DeviceFollowerBatch.from_synth(fb)
        eng = follower.FollowerEngine(enc, dec, store)
        with torch.cuda.stream(s):
            replay, st = eng.capture(batch, 20, 'argmax')
        reps.append((replay, st, batch))
    torch.cuda.synchronize()
    K = 24
    for r, s in zip(reps, streams):
        with torch.cuda.stream(s):
            r[0]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        s = streams[k % NS]
        with torch.cuda.stream(s):
            reps[k % NS][0]()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('streams=%d: %.3f ms per rollout, %.0f agent-steps/s' % (NS, dt / K * 1e3, 2000 * K / dt))
