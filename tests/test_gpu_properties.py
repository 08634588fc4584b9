"""Size-independent properties of the follower path at the HEADLINE sizes (batch 100, 36 views x
2048-d features, <= 80-token instructions, 20 decode steps, a 600-viewpoint table): what must hold
whatever the numbers are, checked where the oracle would take minutes.

 * attention weights are distributions (sum to 1, zero on padding) at every step;
 * a row's results do not depend on which other rows share its batch (rows 0..49 alone == the same
   rows inside the batch of 100), nor on trailing padding of the instruction matrix;
 * permuting a sample's non-stop candidates permutes its logits and changes nothing else;
 * masked (padding) candidates have logit -inf and are never chosen; ended rows stay ended;
 * the loss equals the sum over steps of the mean cross-entropy of the live rows, recomputed from
   the logits the path returns (the reference's definition, follower.py:481, 536-538)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

S, B, NVP = 20, 100, 600


@pytest.fixture(scope='module')
def setup():
    from speaker_follower_amd import synth, model, features, follower
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights(77)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    store = features.FeatureStore(synth.feature_table(77, NVP))
    fb = synth.follower_batch(seed=77, batch=B, steps=S, n_viewpoints=NVP)
    return synth, follower, enc, dec, store, fb


def _run(follower, enc, dec, store, batch, feedback='argmax'):
    eng = follower.FollowerEngine(enc, dec, store)
    with torch.no_grad():
        st = eng.rollout(batch, S, feedback, train=False)
    torch.cuda.synchronize()
    return st


def test_attention_weights_are_distributions(setup):
    synth, follower, enc, dec, store, fb = setup
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    st = _run(follower, enc, dec, store, batch)
    av = st.tape['alpha_v'].cpu().numpy()                 # [S, B, 36]
    at = st.tape['alpha'].cpu().numpy()                   # [S, B, L]
    assert av.shape[:2] == (S, B) and (av >= 0).all() and (at >= 0).all()
    np.testing.assert_allclose(av.sum(-1), 1.0, atol=2e-6)
    np.testing.assert_allclose(at.sum(-1), 1.0, atol=2e-6)
    pad = batch.mask.cpu().numpy().astype(bool)           # [B, L], True = padding
    assert (at[:, pad] == 0).all()


def test_rows_do_not_see_each_other_and_padding_is_inert(setup):
    synth, follower, enc, dec, store, fb = setup
    full = _run(follower, enc, dec, store, follower.DeviceFollowerBatch.from_synth(fb))
    half = _run(follower, enc, dec, store, follower.DeviceFollowerBatch.from_synth(fb, rows=slice(0, 50)))
    assert torch.equal(full.actions[:, :50], half.actions)
    lf, lh = full.logits[:, :50], half.logits
    assert torch.equal(torch.isfinite(lf), torch.isfinite(lh))
    torch.testing.assert_close(torch.nan_to_num(lf, neginf=0.0), torch.nan_to_num(lh, neginf=0.0),
                               rtol=1e-5, atol=1e-5)
    # the shorter half of the batch alone has a narrower instruction matrix (max length of ITS rows)
    tail = _run(follower, enc, dec, store, follower.DeviceFollowerBatch.from_synth(fb, rows=slice(50, 100)))
    assert tail.tape['alpha'].shape[-1] < full.tape['alpha'].shape[-1]
    assert torch.equal(full.actions[:, 50:], tail.actions)
    torch.testing.assert_close(torch.nan_to_num(full.logits[:, 50:], neginf=0.0),
                               torch.nan_to_num(tail.logits, neginf=0.0), rtol=1e-5, atol=1e-5)


def test_permuting_candidates_permutes_logits(setup):
    synth, follower, enc, dec, store, fb = setup
    import copy
    rng = np.random.default_rng(5)
    fb2 = copy.deepcopy(fb)
    perms = np.tile(np.arange(fb.a_max), (S, B, 1))
    for t in range(S):
        for b in range(B):
            n = int(fb.a_num[t, b])
            if n > 2:
                p = 1 + rng.permutation(n - 1)            # candidate 0 (stop) stays where it is
                perms[t, b, 1:n] = p
    ix = (np.arange(S)[:, None, None], np.arange(B)[None, :, None], perms)
    fb2.cand_view = fb.cand_view[ix]
    fb2.cand_heading = fb.cand_heading[ix]
    fb2.cand_elevation = fb.cand_elevation[ix]
    inv = np.argsort(perms, axis=-1)
    fb2.target = np.where(fb.target >= 0, np.take_along_axis(inv, np.maximum(fb.target, 0)[..., None], -1)[..., 0], -1)
    a = _run(follower, enc, dec, store, follower.DeviceFollowerBatch.from_synth(fb), 'teacher')
    b = _run(follower, enc, dec, store, follower.DeviceFollowerBatch.from_synth(fb2), 'teacher')
    la = torch.nan_to_num(a.logits, neginf=-1e30).cpu().numpy()
    lb = torch.nan_to_num(b.logits, neginf=-1e30).cpu().numpy()
    A = la.shape[-1]
    np.testing.assert_allclose(lb, la[ix[0], ix[1], perms[..., :A]], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(float(b.loss), float(a.loss), rtol=1e-5)


def test_masked_candidates_and_ended_rows(setup):
    synth, follower, enc, dec, store, fb = setup
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    st = _run(follower, enc, dec, store, batch)
    lg = st.logits.cpu().numpy()
    act = st.actions.cpu().numpy()
    A = lg.shape[-1]
    valid = np.arange(A)[None, None, :] < fb.a_num[..., None]
    assert np.isneginf(lg[~valid]).all() and np.isfinite(lg[valid]).all()
    assert (act < fb.a_num).all()                          # a padding candidate is never chosen
    ended = np.zeros(B, bool)
    for t in range(S):
        assert (np.argmax(lg[t], -1) == act[t]).all()      # argmax feedback: first maximum
        ended |= act[t] == 0
    assert ended.any()


def test_loss_is_sum_of_per_step_means_over_live_rows(setup):
    synth, follower, enc, dec, store, fb = setup
    st = _run(follower, enc, dec, store, follower.DeviceFollowerBatch.from_synth(fb), 'teacher')
    lg = st.logits.double().cpu()
    tgt = torch.from_numpy(fb.target)
    total = 0.0
    for t in range(S):
        live = tgt[t] >= 0
        if live.any():
            total += float(torch.nn.functional.cross_entropy(lg[t][live], tgt[t][live], reduction='mean'))
    np.testing.assert_allclose(float(st.loss), total, rtol=2e-5)


# ---- speaker path (100 paths of 4-7 steps, 80 word steps) ------------------------------
@pytest.fixture(scope='module')
def speaker_setup():
    from speaker_follower_amd import synth, model, features, speaker
    d = synth.FULL
    enc_w, dec_w = synth.speaker_weights(78)
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=dec_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    sb = synth.speaker_batch(seed=78, batch=B, n_viewpoints=NVP, min_path=4, max_path=7, min_len=10, max_len=79)
    store = features.FeatureStore(synth.feature_table(78, NVP))
    return speaker, enc, dec, store, sb


def test_speaker_teacher_loss_and_scores_follow_their_definition(speaker_setup):
    """speaker.py:172-182: per step log_softmax over the 991 words; loss = sum_t mean over non-PAD
    targets of the NLL; a sample's score = sum over its non-PAD target words of log p."""
    speaker, enc, dec, store, sb = speaker_setup
    W = 80
    eng = speaker.SpeakerEngine(enc, dec, store)
    batch = speaker.DeviceSpeakerBatch.from_synth(sb)
    with torch.no_grad():
        st = eng.score(batch, W, 'teacher', train=False)
    torch.cuda.synchronize()
    lp = torch.log_softmax(st.logits.double(), -1).cpu()       # [W, B, vocab]
    tgt = st.targets.cpu()                                     # [W, B]
    pick = lp.gather(-1, tgt[..., None])[..., 0]
    live = tgt != 0
    loss = sum(float(-pick[t][live[t]].mean()) for t in range(W) if live[t].any())
    np.testing.assert_allclose(float(st.loss), loss, rtol=2e-5)
    np.testing.assert_allclose(st.step_scores.sum(0).cpu().numpy(), (pick * live).sum(0).numpy(),
                               rtol=1e-4, atol=1e-3)
    # the path attention of every word step is a distribution over the path's real steps only
    al = st.tape['alpha'].cpu().numpy()                        # [W, B, Tp]
    np.testing.assert_allclose(al.sum(-1), 1.0, atol=2e-6)
    pad = batch.path_mask.cpu().numpy().astype(bool)           # [B, Tp]
    assert (al[:, pad] == 0).all()


def test_speaker_greedy_words_are_argmax_and_stop_after_eos(speaker_setup):
    speaker, enc, dec, store, sb = speaker_setup
    W = 40
    eng = speaker.SpeakerEngine(enc, dec, store)
    with torch.no_grad():
        st = eng.score(speaker.DeviceSpeakerBatch.from_synth(sb), W, 'argmax', train=False)
    torch.cuda.synchronize()
    words = st.words[1:].cpu().numpy()                         # [W, B]
    am = st.logits.argmax(-1).cpu().numpy()
    ended = np.zeros(B, bool)
    for t in range(W):
        assert (words[t][~ended] == am[t][~ended]).all()       # greedy word = first maximum
        assert (words[t][ended] == 0).all()                    # PAD once EOS was produced (speaker.py:184-191)
        ended |= words[t] == 2
