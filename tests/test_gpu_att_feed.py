"""GPU: SpeakerDecoderLSTM(use_input_att_feed=True) (model.py:475-481, 500-513) on the HIP path -- the module's forward is
a composition of C-ABI operators (ContextOnlySoftDotAttention = Linear + sf_text_attention, LSTMCell, Linear + tanh) under
torch autograd -- against golden G14 (outputs and gradients of the reference's own module over three chained word steps)
and through SpeakerEngine (which steps such a decoder through its module: the fused word loops do not cover it)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from speaker_follower_amd import synth                                # noqa: E402
from test_gpu_hard_parity import check_grads                           # noqa: E402
from tol import assert_logits_close                                    # noqa: E402


def _decoder(seed):
    from speaker_follower_amd import model
    d = synth.FULL
    w = synth.speaker_decoder_att_feed_weights(seed)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=w['embedding.weight'], use_input_att_feed=True)
    dec.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    return dec.cuda()


def test_input_att_feed_module_matches_reference(golden):
    g = golden('g14_speaker_att_feed')
    dec = _decoder(int(g['weight_seed'])).eval()
    assert set(dec.state_dict()) == {'embedding.weight', 'lstm.weight_ih', 'lstm.weight_hh', 'lstm.bias_ih', 'lstm.bias_hh',
                                     'attention_layer.linear_in.weight', 'output_l1.weight', 'output_l1.bias',
                                     'decoder2action.weight', 'decoder2action.bias'}        # model.py:475-485
    dev = lambda a: torch.tensor(a).cuda()                               # noqa: E731
    ctx = dev(g['ctx']).requires_grad_(True)
    h0 = dev(g['h0']).requires_grad_(True)
    c0 = dev(g['c0']).requires_grad_(True)
    mask = dev(g['mask'])
    h, c = h0, c0
    for t in range(3):
        h, c, alpha, logit = dec(dev(g['words'][t]).view(-1, 1), h, c, ctx, mask)
        np.testing.assert_allclose(h.detach().cpu().numpy(), g['h1_%d' % t], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(c.detach().cpu().numpy(), g['c1_%d' % t], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(alpha.cpu().numpy(), g['alpha_%d' % t], rtol=1e-4, atol=2e-6)
        assert_logits_close(logit.detach().cpu().numpy(), g['logit_%d' % t], 'G14 att-feed speaker decoder, word step %d' % t)
    ((logit * dev(g['g_logit'])).sum() + (h * dev(g['g_h'])).sum()).backward()
    check_grads({k: p.grad for k, p in dec.named_parameters() if p.grad is not None}, g, 'dec/')
    for name, t_ in (('d_h0', h0), ('d_c0', c0), ('d_ctx', ctx)):
        want = g[name]
        np.testing.assert_allclose(t_.grad.cpu().numpy(), want, rtol=2e-3, atol=2e-5 * float(np.abs(want).max()), err_msg=name)


def test_input_att_feed_through_the_speaker_engine():
    """SpeakerEngine.score with such a decoder: module-stepped pass == a hand-written loop over the module + the oracle's
    glue arithmetic; teacher loss differentiable (every parameter receives a gradient); train mode draws different masks per
    pass and the same masks for the same site."""
    from speaker_follower_amd import model, features, speaker
    from oracle import np_model                                          # checker
    d = synth.FULL
    senc_w, _ = synth.speaker_weights(31)
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    enc.cuda().eval()
    dec = _decoder(9).eval()
    sb = synth.speaker_batch(seed=3, batch=12, n_viewpoints=48, min_len=4, max_len=9)
    store = features.FeatureStore(synth.feature_table(8, 48))
    batch = speaker.DeviceSpeakerBatch.from_synth(sb)
    eng = speaker.SpeakerEngine(enc, dec, store)
    st = eng.score(batch, 10, 'teacher', train=False)
    assert not st.persistent and not st.teacher_path
    # the oracle's per-step glue on the engine's own logits: words, scores, loss
    lg = st.logits.cpu().numpy()
    tgt = batch.instr_seq[:, :10].t().cpu().numpy()
    loss, ended = 0.0, np.zeros(12, bool)
    for t in range(10):
        terms = np_model.cross_entropy_terms(lg[t], tgt[t], 0)
        n = int((tgt[t] != 0).sum())
        if n:
            loss += float(terms.sum()) / n
        np.testing.assert_allclose(st.step_scores[t].detach().cpu().numpy(), -terms, rtol=1e-4, atol=1e-4)
        ended |= tgt[t] == 2
        if ended.all():
            break
    np.testing.assert_allclose(float(st.loss), loss, rtol=1e-4)
    assert np.array_equal(st.words[1:].cpu().numpy(), tgt)
    st.loss.backward()
    got = {k for k, p in list(enc.named_parameters()) + list(dec.named_parameters()) if p.grad is not None and float(p.grad.abs().max()) > 0}
    assert {'lstm.weight_ih', 'output_l1.weight', 'attention_layer.linear_in.weight', 'decoder2action.weight'} <= got
    # greedy decoding runs too, and train mode is reproducible per site
    with torch.no_grad():
        g1 = eng.score(batch, 10, 'argmax', train=False)
        assert g1.words.shape == (11, 12) and int(g1.words[1:].min()) >= 0
        e2 = speaker.SpeakerEngine(enc, dec, store)
        e2.dropout_seed = 5
        for m in (enc, dec):                       # (module-level dropout: a seed and a call counter per module)
            m._drop_state.seed, m._drop_state.counter = 123, 0
        a = e2.score(batch, 6, 'teacher', train=True).logits.clone()
        b = e2.score(batch, 6, 'teacher', train=True).logits.clone()
        for m in (enc, dec):
            m._drop_state.counter = 0
        c = e2.score(batch, 6, 'teacher', train=True).logits.clone()
    assert not torch.equal(a, b) and torch.equal(a, c)
