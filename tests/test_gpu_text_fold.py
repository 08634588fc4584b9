"""GPU: the FOLDED text attention of inference rollouts (include/sf_hip.h ABI 9: sf_follower_episode.ctx_q / ctx_o;
csrc/sf_attention.hip: text_fold_body; csrc/sf_gemm_small.h: the A-prologue).

model.py:129-141 per decode step: t = W_in h1, s_l = ctx_l . t, alpha = softmax(s), wc = sum alpha_l ctx_l,
h~ = tanh(W_out [wc ; h1]).  The context is constant over an episode, so the engine applies W_in and W_out[:, :H] to it
ONCE and every step scores with ctx_q[l] . h1 and forms h~ = tanh(sum alpha_l ctx_o[l] + W_out[:, H:] h1): the same
function (fp32 re-association), two dependent launches fewer per step.  Checked here: against the unfolded path of the
same build (logits within 3e-5 of scale, identical actions) and against the numpy oracle of the reference (1e-4, actions
bit-exact), on ragged instruction lengths that leave the second position group of a sample empty, at batch 100 and at
a batch that is no multiple of 16; never taken by a rollout that a backward may follow."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speaker_follower_amd import synth                                # noqa: E402
from oracle import np_env, np_model                                   # noqa: E402


def _models(seed=77):
    from speaker_follower_amd import model
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return enc.cuda().eval(), dec.cuda().eval(), enc_w, dec_w


@pytest.mark.parametrize('chain', [True, False])
@pytest.mark.parametrize('B,S,min_len,max_len', [(100, 8, 10, 79), (37, 5, 3, 30), (16, 3, 2, 12)])
def test_folded_text_attention_equals_the_unfolded_path_and_the_oracle(B, S, min_len, max_len, chain):
    """chain: also q' = M_v h1 + c_v and [r | c] = M_a h~ + c_a as single products (sf_follower_episode.chain_fold: three
    dependent launches behind the cell); without: the four-launch folded chain."""
    from speaker_follower_amd import features, follower
    enc, dec, enc_w, dec_w = _models()
    NVP = 96
    fb = synth.follower_batch(seed=11 + B, batch=B, steps=S, n_viewpoints=NVP, min_len=min_len, max_len=max_len)
    table = synth.feature_table(5, NVP)
    store = features.FeatureStore(table)
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    res = {}
    for fold in (True, False):
        eng = follower.FollowerEngine(enc, dec, store)
        eng.fold_text, eng.fold_chain = fold, chain
        with torch.no_grad():
            st = eng.rollout(batch, S, 'argmax', train=False)
        torch.cuda.synchronize()
        assert bool(getattr(st, 'text_folded', False)) == fold
        res[fold] = (st.logits.cpu().numpy().copy(), st.actions.cpu().numpy().copy(), float(st.loss_buf),
                     st.tape['h1'].cpu().numpy().copy(), st.tape['alpha'].cpu().numpy().copy())
    (lf, af, lossf, hf, alf), (lu, au, lossu, hu, alu) = res[True], res[False]
    # the text-attention weights of the tape: distributions, equal to the unfolded ones
    np.testing.assert_allclose(alf.sum(-1), 1.0, atol=1e-5)
    np.testing.assert_allclose(alf, alu, rtol=1e-4, atol=1e-6)
    fin = np.isfinite(lu)
    assert np.array_equal(fin, np.isfinite(lf))
    scale = float(np.abs(lu[fin]).max())
    d = float(np.abs(lf[fin] - lu[fin]).max())
    print('[text fold%s] B=%d S=%d: max |logit| %.2f, folded vs unfolded %.2e, h1 %.2e'
          % (' + chain' if chain else '', B, S, scale, d, np.abs(hf - hu).max()))
    assert 0 < d <= 3e-5 * max(scale, 1.0)                        # (> 0: the folded kernels really ran)
    assert np.array_equal(af, au)
    assert abs(lossf - lossu) <= 1e-5 * max(1.0, abs(lossu))
    # ... and the reference itself (numpy oracle, pinned to the reference's modules by tests/test_oracle_golden.py)
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, 80, reverse=True)
    loc = np_env.static_loc_embeddings()
    ref = np_model.follower_rollout(enc_w, dec_w, seq, lens, mask, S,
                                    lambda t: np_env.dense_follower_step(table, loc, fb, t), fb.target, 'argmax', 2176,
                                    early_exit=False)
    n = len(ref['logits'])
    assert np.array_equal(af[:n], ref['actions'])
    for t in range(n):
        a = ref['logits'][t].shape[1]
        ok = np.isfinite(ref['logits'][t])
        assert float(np.abs(lf[t][:, :a][ok] - ref['logits'][t][ok]).max()) <= 1e-4


def test_a_rollout_that_may_run_backward_is_never_folded():
    from speaker_follower_amd import features, follower
    enc, dec, _, _ = _models()
    fb = synth.follower_batch(seed=3, batch=12, steps=4, n_viewpoints=48, min_len=4, max_len=20)
    store = features.FeatureStore(synth.feature_table(5, 48))
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    eng = follower.FollowerEngine(enc, dec, store)
    st = eng.rollout(batch, 4, 'teacher', train=False)               # eval mode, grad enabled: differentiable
    assert st.differentiable and not st.text_folded
    st.loss.backward()
    assert float(dec.text_attention_layer.linear_in.weight.grad.abs().max()) > 0
    st = eng.rollout(batch, 4, 'teacher', train=True)                # train mode (dropout)
    assert not st.text_folded
    with torch.no_grad():
        st = eng.rollout(batch, 4, 'argmax', train=False)
    assert st.text_folded


def test_captured_rollout_replays_the_folded_chain():
    """The captured inference rollout (what bench.py replays) takes the folded chain; replays after a weight update
    see the new weights in ctx_q / ctx_o (they are rebuilt INSIDE the episode call, i.e. inside the graph)."""
    from speaker_follower_amd import features, follower
    enc, dec, _, _ = _models()
    fb = synth.follower_batch(seed=4, batch=48, steps=5, n_viewpoints=64, min_len=6, max_len=60)
    store = features.FeatureStore(synth.feature_table(5, 64))
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    eng = follower.FollowerEngine(enc, dec, store)
    replay, st = eng.capture(batch, 5, 'argmax')
    assert st.text_folded
    replay()
    torch.cuda.synchronize()
    a = st.logits.clone()
    with torch.no_grad():
        ref = follower.FollowerEngine(enc, dec, store).rollout(batch, 5, 'argmax', train=False)
    assert torch.equal(a, ref.logits)
    with torch.no_grad():
        dec.text_attention_layer.linear_in.weight.mul_(1.5)
    replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        ref2 = follower.FollowerEngine(enc, dec, store).rollout(batch, 5, 'argmax', train=False)
    assert torch.equal(st.logits, ref2.logits) and not torch.equal(st.logits, a)


def test_device_environment_rollout_folded_equals_unfolded():
    """The device-resident environment's schedule (the panorama of step t+1 depends on a_t: attention behind the env
    step) with the folded text stage: same walk, same logits to re-association."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import search_world as W
    from speaker_follower_amd import features, follower, nav
    env, table = W.build_world(dense=False, n_items=24, batch=12, item_seed=77)
    enc, dec, _, _ = _models(303)
    store = features.FeatureStore(table)
    nt = nav.NavTable(env, store)
    env.reset_epoch()
    env._next_minibatch(True)
    items = list(env.batch)
    out = {}
    for fold in (True, False):
        eng = follower.FollowerEngine(enc, dec, store)
        eng.fold_text = fold
        navb = nav.DeviceNavBatch(nt, items, 7)
        with torch.no_grad():
            st = eng.rollout(navb, 7, 'argmax', train=False)
        torch.cuda.synchronize()
        assert bool(st.text_folded) == fold and st.episode is not None
        out[fold] = (st.logits.cpu().numpy().copy(), st.actions.cpu().numpy().copy(), navb.row.cpu().numpy().copy(),
                     navb.view.cpu().numpy().copy())
    (lf, af, rf, vf), (lu, au, ru, vu) = out[True], out[False]
    assert np.array_equal(af, au) and np.array_equal(rf, ru) and np.array_equal(vf, vu)
    fin = np.isfinite(lu)
    assert np.array_equal(fin, np.isfinite(lf))
    d = float(np.abs(lf[fin] - lu[fin]).max())
    assert 0 < d <= 3e-5 * max(1.0, float(np.abs(lu[fin]).max())), d     # (> 0: the folded kernels really ran)


@pytest.mark.parametrize('L', [9, 16, 17, 33, 48, 49, 64, 80])
def test_context_lengths_take_the_right_group_size(L):
    """The position groups are 8 waves x RPW rows (RPW = 1, 2, 3, 5): every context width up to 80, including the widths on
    the boundaries between two templates and rows whose later groups hold only padding."""
    from speaker_follower_amd import features, follower
    enc, dec, _, _ = _models()
    B, S, NVP = 64, 3, 48
    fb = synth.follower_batch(seed=L, batch=B, steps=S, n_viewpoints=NVP, min_len=1, max_len=L - 1)
    fb.instr[0] = np.arange(4, 4 + L - 1, dtype=np.int64)             # (one row of exactly L - 1 tokens + EOS: width L)
    store = features.FeatureStore(synth.feature_table(5, NVP))
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    assert batch.mask.shape[1] == L
    out = {}
    for fold in (True, False):
        eng = follower.FollowerEngine(enc, dec, store)
        eng.fold_text = fold
        with torch.no_grad():
            st = eng.rollout(batch, S, 'argmax', train=False)
        out[fold] = (st.logits.cpu().numpy().copy(), st.actions.cpu().numpy().copy(), st.tape['alpha'].cpu().numpy().copy())
    fin = np.isfinite(out[False][0])
    assert np.array_equal(out[True][1], out[False][1])
    d = float(np.abs(out[True][0][fin] - out[False][0][fin]).max())
    assert 0 < d <= 3e-5 * max(1.0, float(np.abs(out[False][0][fin]).max())), d
    np.testing.assert_allclose(out[True][2], out[False][2], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('B', [1, 17, 49, 112, 113, 192, 193, 224, 256, 257, 300])
def test_every_batch_size_runs_folded_or_falls_back_cleanly(B):
    """The folded chain exists for the batch sizes whose small products take the instantiated plans; every other size must
    take the unfolded stages by itself (never an error, never a half-folded step): same actions either way."""
    from speaker_follower_amd import features, follower
    enc, dec, _, _ = _models()
    S, NVP = 2, 32
    fb = synth.follower_batch(seed=B, batch=B, steps=S, n_viewpoints=NVP, min_len=3, max_len=20)
    store = features.FeatureStore(synth.feature_table(5, NVP))
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    out = {}
    for fold in (True, False):
        eng = follower.FollowerEngine(enc, dec, store)
        eng.fold_text = fold
        with torch.no_grad():
            st = eng.rollout(batch, S, 'argmax', train=False)
        torch.cuda.synchronize()
        out[fold] = (st.logits.cpu().numpy().copy(), st.actions.cpu().numpy().copy())
    fin = np.isfinite(out[False][0])
    assert np.array_equal(fin, np.isfinite(out[True][0])) and np.array_equal(out[True][1], out[False][1])
    assert float(np.abs(out[True][0][fin] - out[False][0][fin]).max()) <= 3e-5 * max(1.0, float(np.abs(out[False][0][fin]).max()))
