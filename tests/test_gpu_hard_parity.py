"""GPU: parity where round 1 was soft (goldens from tests/golden/make_golden_hard.py, i.e. outputs of
the REFERENCE modules run in the build container):

* "peaky" weights -- logits with O(1) spread (std 1.4, max 7.4), visual attention maxima ~0.6, text
  attention maxima ~0.8, yet a contractive recurrence (with 8x visual gains the reference's own fp32
  arithmetic drifts 1e-5 -> 2e-2 over 20 steps between torch and a literal numpy restatement) -- at
  the headline shape, argmax feedback, ALL 20 decode steps pinned for all 100 rows; tolerance
  1e-4 x max|logit| (follower.py:476-505);
* train mode at B = 100: the reference with ITS nn.Dropout replaced by this repo's counter-based
  masks; loss and BPTT gradients (model.py:86-102, 392-395);
* speaker at B = 100: 80-step teacher NLL + gradients, 40 greedy words (speaker.py:158-197);
* `sample` feedback (follower.py:491-497, the default training feedback train.py:299-300): 20 000
  draws per distribution against softmax(logit) by chi-square; invalid candidates never drawn;
  ended rows stay ended and carry no loss.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.tol import assert_logits_close                            # noqa: E402
from speaker_follower_amd import synth                                # noqa: E402


def follower(seed):
    from speaker_follower_amd import model
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return enc.cuda(), dec.cuda()


def check_grads(named, g, prefix, rtol=3e-3):
    seen = 0
    gmax = max(float(v) for k, v in g.items() if k.startswith(prefix + 'gnorm/'))
    for name, grad in named.items():
        key = prefix + 'gnorm/' + name
        if key not in g:
            continue
        seen += 1
        flat = grad.detach().cpu().numpy().ravel()
        norm = np.sqrt(np.sum(flat.astype(np.float64) ** 2))
        if g[key] < 1e-6 * max(gmax, 1.0):     # shift-invariant biases: the true gradient is zero, the
            assert norm < 1e-5 * max(gmax, 1.0), name      # reference holds its own roundoff
            continue
        np.testing.assert_allclose(norm, g[key], rtol=rtol, err_msg=name)
        np.testing.assert_allclose(flat[g[prefix + 'gidx/' + name]], g[prefix + 'gval/' + name],
                                   rtol=rtol, atol=rtol * g[key] / np.sqrt(flat.size) + 1e-7, err_msg=name)
    assert seen > 0


def test_peaky_weights_b100_argmax_all_twenty_steps(golden):
    from speaker_follower_amd import features, follower as fol
    g = golden('g8_follower_peaky_b100_argmax')
    assert int(g['n_steps']) == 20
    enc, dec = follower(int(g['weight_seed']))
    enc.eval()
    dec.eval()
    fb = synth.follower_batch(seed=int(g['batch_seed']), batch=100, steps=20, n_viewpoints=256)
    store = features.FeatureStore(synth.feature_table(int(g['table_seed']), 256))
    batch = fol.DeviceFollowerBatch.from_synth(fb)
    with torch.no_grad():
        st = fol.FollowerEngine(enc, dec, store).rollout(batch, 20, 'argmax', train=False)
    want = g['logits']
    fin = np.isfinite(want)
    scale = float(np.abs(want[fin]).max())
    assert scale > 3.0 and float(want[fin].std()) > 1.0           # the weights really are peaky
    assert float(g['alpha_v'].max(2).mean()) > 0.5                # visual attention has a clear mode
    acts = st.actions.cpu().numpy()
    assert np.array_equal(acts, g['actions'])                     # bit-exact argmax, every step, every row
    got = st.logits.cpu().numpy()
    assert_logits_close(got, want, 'G8 peaky follower B=100, all 20 steps')
    # the attention rows (peaky: a wrong score would move them by orders of magnitude more)
    np.testing.assert_allclose(st.tape['alpha_v'].cpu().numpy(), g['alpha_v'], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(st.tape['alpha'][19].cpu().numpy()[:, :g['alpha_last'].shape[1]], g['alpha_last'],
                               rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(st.h.cpu().numpy(), g['h'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(st.c.cpu().numpy(), g['c'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(float(st.loss), g['loss'], rtol=1e-4)
    np.testing.assert_allclose(st.step_scores.cpu().numpy().sum(0), g['scores'], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('two_stream', [True, False])
def test_peaky_weights_b100_train_mode_loss_and_gradients(golden, two_stream):
    from speaker_follower_amd import features, follower as fol
    g = golden('g8_follower_peaky_b100_train')
    enc, dec = follower(int(g['weight_seed']))
    enc.train()
    dec.train()
    fb = synth.follower_batch(seed=int(g['batch_seed']), batch=100, steps=20, n_viewpoints=256,
                              stop_prob=1.0 / 40.0)
    assert int(g['live_rows'][-1]) >= 50                           # most rows live at step 20
    store = features.FeatureStore(synth.feature_table(int(g['table_seed']), 256))
    batch = fol.DeviceFollowerBatch.from_synth(fb)
    eng = fol.FollowerEngine(enc, dec, store)
    eng.dropout_seed = int(g['dropout_seed'])
    eng.two_stream_backward = two_stream
    st = eng.rollout(batch, 20, 'teacher', train=True)
    assert st.site0 == int(g['site0'])
    want = g['logits']                                             # first 4 steps
    fin = np.isfinite(want)
    got = st.logits[:want.shape[0]].detach().cpu().numpy()
    assert_logits_close(got, want, 'G8 peaky follower B=100, train mode')
    np.testing.assert_allclose(float(st.loss), g['loss'], rtol=1e-4)
    st.loss.backward()
    check_grads({k: p.grad for k, p in enc.named_parameters() if p.grad is not None}, g, 'enc/')
    check_grads({k: p.grad for k, p in dec.named_parameters() if p.grad is not None}, g, 'dec/')


@pytest.mark.parametrize('gate_product', ['bf16x6', 'fp32'])
@pytest.mark.parametrize('feedback', ['teacher', 'argmax'])
def test_speaker_b100_golden(golden, feedback, gate_product):
    """G9: the speaker at B = 100 on the reference's peaky weights (attention scores up to +-80, |logit| up to 17).

    north_star: logits within 1e-4 ABSOLUTE of the reference.  The reference's own fp32 output lies 2.3e-4 (word step
    0) / 0.98e-4 (teacher, step 79) / 1.36e-4 (argmax, step 39) from the SAME modules evaluated in float64
    (tests/golden/make_golden_f64.py): at this scale an fp32 evaluation of the path encoder's attention chain loses
    2e-6 .. 9e-6 per stage in the softmax weights and the 7-step context carries it into every word step
    (tools/speaker_drift.py).  Rounds 1-4 ran the same chain in fp32 and sat at 1.5e-4 .. 2.1e-4; since round 5 the
    encoder's query and scores are float64 (csrc/sf_precise.hip) and the bound is asserted WITHOUT widening:
      * every word step (8 vocabulary columns of all rows), the first and the last step (all columns): within 1e-4
        absolute of the float64 anchor -- measured 5e-5;
      * never further from exact arithmetic than the reference's own fp32 run is, at the first and the last step;
      * against the fp32 golden itself only what the triangle inequality allows: 1e-4 + the golden's own distance.
    The default gate product (bf16x6 split) is held to all three.  The STRICT order (fp32 MFMA gate product,
    runtime.strict_gate_product: 1 216 fp32 roundings per gate where the split form has 152) is the less accurate
    kernel -- 2.9e-7 against 1.1e-7 per step in h (tools/speaker_drift.py) -- and is held to 1e-4 at word step 0 and
    along the pass, and at the last step to the triangle bound only (measured 0.5e-4 teacher / 1.4e-4 argmax)."""
    from speaker_follower_amd import model, features, speaker, _lib
    g = golden('g9_speaker_b100_' + feedback)
    d = synth.FULL
    senc_w, sdec_w = synth.speaker_weights_peaky(int(g['weight_seed']))
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    sb = synth.speaker_batch(seed=int(g['batch_seed']), batch=100, n_viewpoints=256, min_len=10, max_len=79)
    store = features.FeatureStore(synth.feature_table(int(g['table_seed']), 256))
    batch = speaker.DeviceSpeakerBatch.from_synth(sb)
    n = int(g['n_steps'])
    _lib.lib.sf_debug_gate_product_f32(1 if gate_product == 'fp32' else 0)
    try:
        with torch.set_grad_enabled(feedback == 'teacher'):
            st = speaker.SpeakerEngine(enc, dec, store).score(batch, n, feedback, train=False)
        torch.cuda.synchronize()
    finally:
        _lib.lib.sf_debug_gate_product_f32(0)
    np.testing.assert_array_equal(st.words[1:].cpu().numpy(), g['words'])        # bit-exact greedy words
    lg = st.logits.detach().cpu().numpy()
    scale = float(np.abs(g['logit_last']).max())
    assert scale > 5.0
    f64 = golden('g9_speaker_b100_f64')
    tag = 'G9 speaker B=100 %s (%s gate product)' % (feedback, gate_product)
    cols = f64['cols']
    assert_logits_close(lg[:, :, cols], f64[feedback + '/logits_cols'], tag + ', EVERY word step at 8 columns, vs float64')
    for name, got, key in (('word step 0', lg[0], 'first'), ('word step %d' % (n - 1), lg[n - 1], 'last')):
        anchor = f64['%s/%s' % (feedback, 'logits_first' if key == 'first' else 'logit_last')]
        ref32 = g['logits_first'][0] if key == 'first' else g['logit_last']
        own = float(f64['%s/ref32_dist_%s' % (feedback, key)])
        strict_last = gate_product == 'fp32' and key == 'last'
        dist = assert_logits_close(got, anchor, '%s, %s, vs the reference in float64' % (tag, name),
                                   atol=1e-4 + own if strict_last else 1e-4)
        d32 = float(np.abs(got - ref32).max())
        print('[parity]   ... vs the fp32 golden: %.3e; the fp32 golden is itself %.3e from its float64 evaluation' % (d32, own))
        if not strict_last:
            assert dist <= own, 'further from exact arithmetic (%.3e) than the reference\'s own fp32 run (%.3e)' % (dist, own)
        assert d32 <= 1e-4 + own + (own if strict_last else 0.0)
    if 'ctx_rows4' in g:
        np.testing.assert_allclose(st.ctx.detach().cpu().numpy()[::4], g['ctx_rows4'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(float(st.loss), g['loss'], rtol=1e-4)
    np.testing.assert_allclose(st.step_scores.sum(0).detach().cpu().numpy(), g['scores'], rtol=1e-4, atol=2e-3)
    if feedback == 'teacher':
        st.loss.backward()
        check_grads({k: p.grad for k, p in enc.named_parameters() if p.grad is not None}, g, 'enc/')
        check_grads({k: p.grad for k, p in dec.named_parameters() if p.grad is not None}, g, 'dec/')


@pytest.mark.parametrize('case', ['peaky', 'flat', 'two'])
def test_sample_feedback_draws_from_the_masked_softmax(case):
    """follower.py:491-497: probs = softmax(logit); probs[is_valid == 0] = 0; Categorical(probs).sample().
    N rows share one logit vector; each row draws from its own counter-based stream."""
    from scipy import stats
    from speaker_follower_amd import _lib
    from speaker_follower_amd.runtime import ptr, stream
    N, A, F = 20000, 9, 8
    rng = np.random.default_rng(5)
    a_num = dict(peaky=7, flat=9, two=2)[case]
    base = dict(peaky=rng.standard_normal(A) * 2.5, flat=rng.standard_normal(A) * 0.05,
                two=np.array([0.3, -0.4] + [0.0] * (A - 2)))[case].astype(np.float32)
    valid = (np.arange(A) < a_num).astype(np.float32)
    logit = torch.tensor(np.tile(base, (N, 1))).cuda().contiguous()
    is_valid = torch.tensor(np.tile(valid, (N, 1))).cuda().contiguous()
    U = torch.zeros(N, A, F, device='cuda')
    ended_in = (np.arange(N) % 10 == 0).astype(np.uint8)           # every 10th row ended before this step
    ended = torch.tensor(ended_in).cuda()
    target = torch.ones(N, dtype=torch.int64, device='cuda')
    a_t = torch.full((N,), -7, dtype=torch.int64, device='cuda')
    tused = torch.empty(N, dtype=torch.int64, device='cuda')
    score, ce, live = (torch.empty(N, device='cuda') for _ in range(3))
    cands = _lib.Cands(U.data_ptr(), None, None, None, None, None, A, 1, F, 0)
    glue = _lib.FollowerGlue(is_valid.data_ptr(), target.data_ptr(), 2, ended.data_ptr(), a_t.data_ptr(),
                             tused.data_ptr(), score.data_ptr(), None, 0, None, 0, ce.data_ptr(),
                             live.data_ptr(), 0xC0FFEE, 11, 0)
    _lib.call('sf_follower_glue_fwd', C.byref(cands), N, ptr(logit), C.byref(glue), stream())
    torch.cuda.synchronize()
    a = a_t.cpu().numpy()
    assert a.min() >= 0 and a.max() < a_num                        # an invalid candidate is never drawn
    p = np.exp(base[:a_num].astype(np.float64) - base[:a_num].max())
    p /= p.sum()
    counts = np.bincount(a, minlength=a_num)[:a_num]
    chi2, pval = stats.chisquare(counts, p * N)
    assert pval > 1e-4, (case, counts.tolist(), (p * N).round(1).tolist(), chi2, pval)
    # consecutive rows are independent draws: lag-1 agreement matches sum p^2
    agree = float(np.mean(a[1:] == a[:-1]))
    assert abs(agree - float(np.sum(p * p))) < 5 * np.sqrt(1.0 / N)
    # a different stream id gives a different (equally distributed) sample
    glue.sample_stream = 12
    a2 = torch.empty_like(a_t)
    glue.a_t = a2.data_ptr()
    ended.copy_(torch.tensor(ended_in).cuda())
    logit.copy_(torch.tensor(np.tile(base, (N, 1))).cuda())
    _lib.call('sf_follower_glue_fwd', C.byref(cands), N, ptr(logit), C.byref(glue), stream())
    torch.cuda.synchronize()
    assert float(np.mean(a2.cpu().numpy() == a)) < float(np.sum(p * p)) + 0.05
    # bookkeeping: rows ended before the step carry no loss and stay ended; live rows end iff they drew stop
    e = ended.cpu().numpy()
    lv, cet = live.cpu().numpy(), ce.cpu().numpy()
    a2n = a2.cpu().numpy()
    assert np.all(e[ended_in == 1] == 1) and np.all(lv[ended_in == 1] == 0) and np.all(cet[ended_in == 1] == 0)
    assert np.array_equal(e[ended_in == 0] == 1, a2n[ended_in == 0] == 0)
    assert np.all(lv[ended_in == 0] == 1)
    # score = log p(a_t) of the drawn action (follower.py:504)
    np.testing.assert_allclose(score.cpu().numpy(), np.log(p)[a2n].astype(np.float32), rtol=1e-4, atol=1e-5)
