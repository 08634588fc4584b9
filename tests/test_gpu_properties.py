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
