"""GPU: the float64 query / score path of the speaker's path encoder (csrc/sf_precise.hip, round 5).

`sf_linear_f64` (v_mfma_f64_16x16x4_f64) against numpy float64 to float64 roundoff, with asymmetric operands;
`sf_visual_attention_fwd_f64` against a float64 evaluation of VisualSoftDotAttention (model.py:310-326) on inputs whose
attention scores reach +-60: softmax weights within 1e-6 where the fp32 kernels (and any fp32 evaluation) sit at 1e-5."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speaker_follower_amd import synth                                # noqa: E402


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('M,N,K', [(100, 256, 512), (100, 2176, 256), (7, 48, 20), (33, 17, 132), (1, 16, 4)])
def test_linear_on_the_float64_matrix_cores(M, N, K):
    from speaker_follower_amd import _lib
    from speaker_follower_amd.runtime import ptr, stream
    rng = np.random.default_rng(M + 3 * N + K)
    # asymmetric, badly scaled operands: a swapped fragment mapping or an fp32 accumulation cannot pass
    x = (rng.standard_normal((M, K)) * (1.0 + 50.0 * (np.arange(K) % 5 == 0))).astype(np.float32)
    w = (rng.standard_normal((N, K)) * (1.0 + (np.arange(N) % 3)[:, None])).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    y64 = torch.empty(M, N, dtype=torch.float64, device='cuda')
    y32 = torch.empty(M, N, dtype=torch.float32, device='cuda')
    xd, wd, bd = dev(x), dev(w), dev(b)
    _lib.call('sf_linear_f64', ptr(xd), K, ptr(wd), K, ptr(bd), M, N, K, C.c_void_p(y64.data_ptr()), ptr(y32), stream())
    torch.cuda.synchronize()
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    scale = float(np.abs(ref).max())
    err = float(np.abs(y64.cpu().numpy() - ref).max()) / scale
    print('[linear_f64 %dx%dx%d] max error / scale %.2e' % (M, N, K, err))
    assert err <= 1e-14
    assert np.array_equal(y32.cpu().numpy(), y64.cpu().numpy().astype(np.float32))      # the fp32 copy is the rounded f64


@pytest.mark.parametrize('B,indexed', [(100, True), (37, False), (3, True)])
def test_visual_attention_float64_scores(B, indexed):
    from speaker_follower_amd import _lib, features
    from speaker_follower_amd.runtime import ptr, ws_args, transposed
    d = synth.FULL
    H, F, D, V, NVP = d.hidden, d.feat, 256, 36, 64
    rng = np.random.default_rng(B)
    table = synth.feature_table(3, NVP)
    store = features.FeatureStore(table)
    vp = rng.integers(0, NVP, B).astype(np.int32)
    view = rng.integers(0, 36, B).astype(np.int32)
    from oracle import np_env                                          # checker
    loc = np_env.static_loc_embeddings()
    X = np.concatenate([table[vp], loc[view]], axis=2).astype(np.float32)               # [B,36,F]
    # weights scaled until the scores are as large as the peaky speaker's (|score| ~ 60)
    w_h = (rng.standard_normal((D, H)) * 0.25).astype(np.float32)
    b_h = rng.standard_normal(D).astype(np.float32)
    w_v = (rng.standard_normal((D, F)) * 0.015).astype(np.float32)
    b_v = rng.standard_normal(D).astype(np.float32)
    h = np.tanh(rng.standard_normal((B, H))).astype(np.float32)
    t64 = h.astype(np.float64) @ w_h.astype(np.float64).T + b_h
    q64 = t64 @ w_v.astype(np.float64)
    s64 = np.einsum('bvf,bf->bv', X.astype(np.float64), q64)
    e = np.exp(s64 - s64.max(1, keepdims=True))
    a64 = e / e.sum(1, keepdims=True)
    out64 = np.einsum('bv,bvf->bf', a64, X.astype(np.float64))
    assert np.abs(s64).max() > 25.0
    tw = [dev(a) for a in (w_h, b_h, w_v, b_v)]
    vw = _lib.VisualW(*(t.data_ptr() for t in tw), transposed(tw[2]).data_ptr(), transposed(tw[0]).data_ptr())
    vp_d, view_d = dev(vp), dev(view)
    if indexed:
        pano = store.pano(vp_d, view_d)
    else:
        Xd = dev(X)
        pano = _lib.Pano(Xd.data_ptr(), None, None, None, None, V, F, 0)
    errs = {}
    hd = dev(h)
    for name in ('sf_visual_attention_fwd', 'sf_visual_attention_fwd_f64'):
        out = torch.empty(B, F, device='cuda')
        alpha = torch.empty(B, V, device='cuda')
        t_v, q = torch.empty(B, D, device='cuda'), torch.empty(B, F, device='cuda')
        _lib.call(name, C.byref(vw), C.byref(pano), B, H, D, ptr(hd), ptr(out), F, ptr(alpha), ptr(t_v), ptr(q),
                  None, 0, 0, *ws_args(torch.device('cuda', 0)))
        torch.cuda.synchronize()
        errs[name] = (float(np.abs(alpha.cpu().numpy() - a64).max()), float(np.abs(out.cpu().numpy() - out64).max()))
        if name.endswith('f64'):
            assert np.array_equal(t_v.cpu().numpy(), t64.astype(np.float32))           # rounded once
            np.testing.assert_allclose(q.cpu().numpy(), q64, rtol=2e-7, atol=1e-30)
    print('[visual attention, max|score| %.1f] alpha / out error: fp32 kernels %.2e / %.2e, float64 scores %.2e / %.2e'
          % ((np.abs(s64).max(),) + errs['sf_visual_attention_fwd'] + errs['sf_visual_attention_fwd_f64']))
    assert errs['sf_visual_attention_fwd_f64'][0] <= 1e-6
    assert errs['sf_visual_attention_fwd_f64'][1] <= 5e-6 * max(1.0, float(np.abs(out64).max()))
    assert errs['sf_visual_attention_fwd'][0] <= 1e-3                                   # (sanity of the comparison)


def test_folded_attention_query_of_the_speaker_encoder():
    """Inference through the float64 fold M_v = W_v^T W_h (sf_visual_query_fold_f64, sf_speaker_encoder_fwd_folded): the
    same words, logits within 2e-5 of the two-product float64 path (both are within 3e-5 of exact arithmetic, G9), the
    fold follows a weight update in place, and a pass that will be differentiated does not use it."""
    from speaker_follower_amd import model, features, speaker, synth
    d = synth.FULL
    w_enc, w_dec = synth.speaker_weights(23)
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=w_dec['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in w_enc.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in w_dec.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    store = features.FeatureStore(synth.feature_table(5, 64))
    sb = synth.speaker_batch(seed=3, batch=24, n_viewpoints=64, min_path=3, max_path=7, min_len=8, max_len=30)
    batch = speaker.DeviceSpeakerBatch.from_synth(sb)
    eng = speaker.SpeakerEngine(enc, dec, store)
    out = {}
    with torch.no_grad():
        for fold in (True, False):
            eng.fold_query = fold
            for fb in ('teacher', 'argmax'):
                st = eng.score(batch, 30, fb, train=False)
                torch.cuda.synchronize()
                out[fold, fb] = (st.words.clone(), st.logits.clone(), st.e['q'].clone())
        for fb in ('teacher', 'argmax'):
            assert torch.equal(out[True, fb][0], out[False, fb][0])
            dl = float((out[True, fb][1] - out[False, fb][1]).abs().max())
            dq = float((out[True, fb][2] - out[False, fb][2]).abs().max() / out[False, fb][2].abs().max())
            print('[query fold, %s] max |logit difference| %.2e, query difference / scale %.2e' % (fb, dl, dq))
            assert dl < 2e-5 and dq < 2e-7
        # a weight update: the fold is rebuilt in place (same buffers) and follows it
        eng.fold_query = True
        fold0 = eng.score(batch, 30, 'teacher', train=False).e['fold']
        ptrs = [t.data_ptr() for t in fold0]
        before = fold0[0].clone()
        enc.visual_attention_layer.linear_in_h.weight.mul_(1.01)
        st = eng.score(batch, 30, 'teacher', train=False)
        assert [t.data_ptr() for t in st.e['fold']] == ptrs and not torch.equal(st.e['fold'][0], before)
        eng.fold_query = False
        ref = eng.score(batch, 30, 'teacher', train=False)
        assert float((st.logits - ref.logits).abs().max()) < 2e-5
    eng.fold_query = True
    st = eng.score(batch, 30, 'teacher', train=False)            # grad enabled, trainable weights: the tape path
    assert 'fold' not in st.e and st.loss.requires_grad
