"""GPU: the teacher-forced speaker pass in its batched form (sf_speaker_teacher_fwd / _bwd, round 5) against the word
loop issued step by step (sf_speaker_words_fwd / _bwd) -- the same function, element by element:

  recurrence alone as ONE persistent launch (the encoder's recurrence kernel with a given initial state) + dropout(h1),
  attention, h~, vocabulary projection and the glue for all S*B rows at once; the backward likewise.

Eval and train mode (dropout masks: the per-step sites), ragged path and instruction lengths, B below / at the
persistent kernels' row-group size; and against the golden G9 (B = 100, 80 words) in tests/test_gpu_hard_parity.py, whose
teacher case runs this path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speaker_follower_amd import synth                                # noqa: E402


def _setup(B, seed=404, train=False):
    from speaker_follower_amd import model, features, speaker
    d = synth.FULL
    senc_w, sdec_w = synth.speaker_weights_peaky(seed)
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    enc.cuda().train(train)
    dec.cuda().train(train)
    sb = synth.speaker_batch(seed=B + 1, batch=B, n_viewpoints=96, min_len=5, max_len=40)
    store = features.FeatureStore(synth.feature_table(8, 96))
    return enc, dec, store, speaker.DeviceSpeakerBatch.from_synth(sb)


@pytest.mark.parametrize('B,S,train', [(100, 41, False), (100, 41, True), (37, 24, True), (128, 12, False), (1, 7, True)])
def test_batched_teacher_pass_equals_the_word_loop(B, S, train):
    from speaker_follower_amd import speaker
    out = {}
    for batched in (False, True):
        enc, dec, store, batch = _setup(B, train=train)
        eng = speaker.SpeakerEngine(enc, dec, store)
        eng.dropout_seed = 1234
        eng.teacher_batched = batched
        st = eng.score(batch, S, 'teacher', train=train)
        assert st.teacher_path == batched
        st.loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.clone() for k, p in list(enc.named_parameters()) + list(dec.named_parameters()) if p.grad is not None}
        out[batched] = dict(logits=st.logits.detach().clone(), h1=st.tape['h1'].clone(), c1=st.tape['c1'].clone(),
                            alpha=st.tape['alpha'].clone(), ht=st.tape['h_tilde'].clone(), words=st.words.clone(),
                            scores=st.step_scores.clone(), loss=float(st.loss.detach()), ended=st.ended.clone(),
                            live=st.live.clone(), grads=grads)
    a, b = out[False], out[True]
    assert torch.equal(a['words'], b['words']) and torch.equal(a['ended'], b['ended']) and torch.equal(a['live'], b['live'])
    for k, tol in (('h1', 3e-6), ('c1', 6e-6), ('alpha', 2e-5), ('ht', 2e-5)):
        d = float((a[k] - b[k]).abs().max())
        assert d <= tol, (k, d)
    dl = float((a['logits'] - b['logits']).abs().max())
    scale = float(a['logits'].abs().max())
    print('[teacher batched vs word loop, B=%d S=%d train=%s] max |dlogit| %.2e at max |logit| %.1f, loss %.6f / %.6f'
          % (B, S, train, dl, scale, a['loss'], b['loss']))
    # two fp32 evaluations (split-product persistent recurrence vs per-step fp32 kernels): 1e-4 at |logit| <= 16; train mode
    # doubles the surviving activations (dropout 0.5) and with them the logits
    assert dl <= 1e-4 * max(1.0, scale / 16.0)
    np.testing.assert_allclose(b['scores'].cpu().numpy(), a['scores'].cpu().numpy(), rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(b['loss'], a['loss'], rtol=2e-5)
    assert set(a['grads']) == set(b['grads']) and len(a['grads']) >= 12
    for k in a['grads']:
        ga, gb = a['grads'][k], b['grads'][k]
        s = float(ga.abs().max())
        assert torch.isfinite(gb).all(), k
        assert float((ga - gb).abs().max()) <= 2e-4 * max(s, 1e-3), (k, float((ga - gb).abs().max()), s)


def test_batched_teacher_inference_equals_the_persistent_word_loop():
    """No-grad teacher scoring (the rescoring pass of configs[4]): batched form == sf_speaker_decode."""
    from speaker_follower_amd import speaker
    enc, dec, store, batch = _setup(100)
    res = {}
    with torch.no_grad():
        for batched in (False, True):
            eng = speaker.SpeakerEngine(enc, dec, store)
            eng.teacher_batched = batched
            st = eng.score(batch, 41, 'teacher', train=False)
            torch.cuda.synchronize()
            assert st.teacher_path == batched and st.persistent == (not batched)
            res[batched] = (st.logits.clone(), st.step_scores.clone(), float(st.loss), st.words.clone())
    assert torch.equal(res[False][3], res[True][3])
    assert float((res[False][0] - res[True][0]).abs().max()) <= 1e-4
    np.testing.assert_allclose(res[True][1].cpu().numpy(), res[False][1].cpu().numpy(), rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(res[True][2], res[False][2], rtol=2e-5)


def test_teacher_pass_falls_back_to_the_word_loop_when_its_recurrence_is_starved():
    """The batched form runs its recurrence as a persistent launch: under a forced timeout SpeakerEngine.run re-issues
    the pass on the per-step kernels (same sites) -- forward values of an undisturbed pass, `fallbacks` counted."""
    from speaker_follower_amd import speaker, _lib, runtime
    enc, dec, store, batch = _setup(40)
    dev = torch.device('cuda', 0)
    with torch.no_grad():
        good = speaker.SpeakerEngine(enc, dec, store).score(batch, 20, 'teacher', train=False)
        torch.cuda.synchronize()
        assert good.teacher_path
        runtime.take_fault(dev)
        _lib.lib.sf_debug_persist_timeout(0)
        try:
            eng = speaker.SpeakerEngine(enc, dec, store)
            raw = eng.score(batch, 20, 'teacher', train=False)
            torch.cuda.synchronize()
            assert raw.teacher_path and torch.isnan(raw.logits).any()          # poisoned ...
            assert runtime.take_fault(dev) != 0                                # ... and visible to the host
            st = eng.run(batch, 20, 'teacher', train=False)
        finally:
            _lib.lib.sf_debug_persist_timeout(-1)
            torch.cuda.synchronize()
            runtime.take_fault(dev)
    assert eng.fallbacks == 1 and not st.teacher_path and not st.persistent
    assert torch.equal(st.words, good.words)
    assert float((st.logits - good.logits).abs().max()) <= 1e-4
