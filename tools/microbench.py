#!/usr/bin/env python3
"""Per-operator timings of the C-ABI entry points at the headline shape (B=100), with HIP
events on torch's current stream.  Development aid, not part of the product."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_follower_amd import synth, model, features, follower, ops, _lib  # noqa: E402
from speaker_follower_amd.runtime import ptr, ws_args, stream                   # noqa: E402


def timeit(name, fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    t_host = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    print('%-44s %9.2f us/call (gpu)  %9.2f us/call (host issue)' % (name, e0.elapsed_time(e1) * 1e3 / reps, t_host))


def main():
    B = int(os.environ.get('B', 100))
    dev = torch.device('cuda')
    d = synth.FULL
    H, F, D, V = d.hidden, d.feat, d.dot, d.views
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    for (M, N, K) in [(B, 256, 512), (B, 512, 512), (B, 512, 1024), (B, 2176, 256), (B, 2048, 4864),
                      (B, 2048, 512), (B, 991, 512), (80 * B, 2048, 300)]:
        x, w, b = r(M, K), r(N, K), r(N)
        timeit('linear_fwd M=%d N=%d K=%d' % (M, N, K), lambda: ops.linear_fwd(x, w, b))
    w4 = [r(4 * H, 2 * F), r(4 * H, H), r(4 * H), r(4 * H)]
    x, h, c = r(B, 2 * F), r(B, H), r(B, H)
    timeit('lstm_cell_fwd I=4352', lambda: ops.lstm_cell_fwd(w4, x, h, c))
    w4s = [r(4 * H, 300), r(4 * H, H), r(4 * H), r(4 * H)]
    xs = r(B, 300)
    timeit('lstm_cell_fwd I=300 (fused step)', lambda: ops.lstm_cell_fwd(w4s, xs, h, c))
    table = torch.rand(512, 36, 2048, device=dev)
    store = features.FeatureStore(table)
    vp = torch.randint(0, 512, (B,), device=dev, dtype=torch.int32)
    view = torch.randint(0, 36, (B,), device=dev, dtype=torch.int32)
    pano = store.pano(vp, view)
    wv = [r(D, H), r(D), r(D, F), r(D)]
    timeit('visual_attention_fwd (indexed)', lambda: ops.visual_attention_fwd(wv, pano, B, V, F, h))
    ctx = r(B, 80, H)
    mask = torch.zeros(B, 80, dtype=torch.uint8, device=dev)
    w2 = (r(H, H), r(H, 2 * H))
    timeit('soft_dot_attention_fwd L=80', lambda: ops.soft_dot_attention_fwd(w2, h, ctx, mask))
    A = 14
    cv = torch.randint(0, 36, (B, A), device=dev, dtype=torch.int32)
    sc = r(B, A, 4)
    an = torch.randint(2, A + 1, (B,), device=dev, dtype=torch.int32)
    cnd = store.cands(vp, cv, sc, an, A)
    w6 = [r(D, H), r(D), r(D, F), r(D), r(1, D), r(1)]
    timeit('eltwise_prod_scoring_fwd A=14', lambda: ops.eltwise_prod_scoring_fwd(w6, cnd, B, A, F, h))
    # encoder + full rollout through the engine
    enc_w, dec_w = synth.follower_weights(1)
    enc = model.EncoderLSTM(d.vocab, d.word, H, 0, 0.5, glove=enc_w['embedding.weight']).cuda().eval()
    dec = model.AttnDecoderLSTM(F, H, 0.5, feature_size=F).cuda().eval()
    fb = synth.follower_batch(seed=0, batch=B, steps=20, n_viewpoints=512)
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    with torch.no_grad():
        timeit('EncoderLSTM module fwd (T=%d)' % max(batch.lengths),
               lambda: enc(batch.seq, batch.lengths), reps=10, warm=2)
        eng = follower.FollowerEngine(enc, dec, store)
        timeit('engine.rollout 20 steps', lambda: eng.rollout(batch, 20, 'argmax'), reps=10, warm=2)
        timeit('engine.rollout 1 step', lambda: eng.rollout(batch, 1, 'argmax'), reps=10, warm=2)




def graph_test():
    """hipGraph replay of the whole 20-step rollout vs eager issue."""
    B = 100
    dev = torch.device('cuda')
    d = synth.FULL
    H, F = d.hidden, d.feat
    table = torch.rand(512, 36, 2048, device=dev)
    store = features.FeatureStore(table)
    enc_w, dec_w = synth.follower_weights(1)
    enc = model.EncoderLSTM(d.vocab, d.word, H, 0, 0.5, glove=enc_w['embedding.weight']).cuda().eval()
    dec = model.AttnDecoderLSTM(F, H, 0.5, feature_size=F).cuda().eval()
    fb = synth.follower_batch(seed=0, batch=B, steps=20, n_viewpoints=512)
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    eng = follower.FollowerEngine(enc, dec, store)
    with torch.no_grad():
        for _ in range(3):
            st = eng.rollout(batch, 20, 'argmax')
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            eng.rollout(batch, 20, 'argmax')
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                st2 = eng.rollout(batch, 20, 'argmax')
        torch.cuda.synchronize()
        timeit('graph replay: rollout 20 steps', g.replay, reps=20, warm=3)
        print('actions equal eager:', bool(torch.equal(st.actions, st2.actions)), 'loss', float(st2.loss))


if os.environ.get('GRAPH'):
    graph_test()

if __name__ == '__main__' and not os.environ.get('GRAPH'):
    main()
