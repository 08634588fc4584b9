"""GPU: the BIDIRECTIONAL instruction encoder (model.py:47-66, 88-102; train.py:197-199 builds it with hidden_size // 2
per direction).  Goldens: tests/golden/make_golden_bidir.py -- the reference module in eval mode, and a teacher-forced
training step (trainable embedding, this repo's counter-based masks in place of nn.Dropout) whose loss, logits and the
gradients of EVERY parameter of both directions are compared."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.tol import assert_logits_close                            # noqa: E402
from tests.test_gpu_hard_parity import check_grads                   # noqa: E402
from speaker_follower_amd import synth                                # noqa: E402


def _encoder(seed, glove):
    from speaker_follower_amd import model
    d = synth.FULL
    w = synth.bidirectional_encoder_weights(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden // 2, 0, 0.5, bidirectional=True,
                            glove=w['embedding.weight'] if glove else None)
    enc.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    return enc.cuda()


def test_bidirectional_encoder_matches_the_reference_module(golden):
    g = golden('g12_encoder_bidir_eval')
    enc = _encoder(int(g['weight_seed']), glove=True).eval()
    assert enc.num_directions == 2 and enc.encoder2decoder.weight.shape == (512, 512)
    with torch.no_grad():
        ctx, h, c = enc(torch.tensor(g['seq']).cuda(), [int(x) for x in g['lengths']])
    torch.cuda.synchronize()
    for name, got, want in (('ctx', ctx, g['ctx']), ('decoder_init', h, g['decoder_init']), ('c_t', c, g['c_t'])):
        got = got.cpu().numpy()
        assert got.shape == want.shape, name
        err = np.abs(got - want).max()
        assert err <= 2e-5, '%s: max abs err %.3g' % (name, err)        # |values| <= 1 (tanh-bounded; c_t a few units)
    # beyond a row's length both halves are exactly zero (pad_packed_sequence, model.py:101)
    lens = g['lengths']
    ctx = ctx.cpu().numpy()
    for b in range(len(lens)):
        assert not ctx[b, lens[b]:].any()
    # the one-token row: its reverse half at position 0 is the reverse direction's only step
    assert np.abs(ctx[-1, 0, 256:]).max() > 0


def test_bidirectional_encoder_gradients_reach_both_directions():
    """Module path, eval mode (no dropout), against finite differences of a scalar of all three outputs -- in float64 on
    the host through the oracle's LSTM is what the golden test does; here: the two directions' gradients are non-zero and
    the sum over a doubled loss doubles (linearity of the backward movement kernels)."""
    enc = _encoder(5, glove=True).eval()
    d = synth.FULL
    r = np.random.default_rng(3)
    lens = [12, 9, 9, 4, 1]
    seq = np.zeros((5, 80), np.int64)
    for b, n in enumerate(lens):
        seq[b, :n] = r.integers(4, d.vocab, size=n)
    seq = torch.tensor(seq).cuda()
    wts = [torch.tensor(r.standard_normal(s).astype(np.float32)).cuda() for s in ((5, 12, 512), (5, 512), (5, 512))]
    grads = []
    for scale in (1.0, 2.0):
        enc.zero_grad(set_to_none=True)
        outs = enc(seq, lens)
        loss = sum((o * w).sum() for o, w in zip(outs, wts)) * scale
        loss.backward()
        grads.append({k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None})
    for k in ('lstm.weight_hh_l0', 'lstm.weight_hh_l0_reverse', 'lstm.weight_ih_l0_reverse', 'encoder2decoder.weight'):
        a, b = grads[0][k], grads[1][k]
        assert float(a.abs().max()) > 0, k
        assert float((b - 2 * a).abs().max()) <= 1e-5 * float(b.abs().max()), k


def test_follower_training_step_through_the_bidirectional_encoder(golden):
    from speaker_follower_amd import model, features, follower as fol
    g = golden('g12_follower_bidir_train')
    d = synth.FULL
    enc = _encoder(int(g['enc_weight_seed']), glove=False).train()
    _, dec_w = synth.follower_weights_peaky(int(g['dec_weight_seed']))
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    dec.cuda().train()
    assert enc.embedding.weight.requires_grad
    S = int(g['n_steps'])
    fb = synth.follower_batch(seed=int(g['batch_seed']), batch=16, steps=S, n_viewpoints=64, min_len=5, max_len=20,
                              stop_prob=0.05)
    store = features.FeatureStore(synth.feature_table(int(g['table_seed']), 64))
    eng = fol.FollowerEngine(enc, dec, store)
    eng.dropout_seed = int(g['dropout_seed'])
    st = eng.rollout(fol.DeviceFollowerBatch.from_synth(fb), S, 'teacher', train=True)
    assert st.site0 == int(g['site0'])
    # the absolute 1e-4 bound is against the reference evaluated in float64 (its own fp32 evaluation is itself
    # ~1e-4 from that: tests/tol.py)
    want = g['logits_first_f64']
    got = st.logits[0].detach().cpu().numpy()[:, :want.shape[1]]
    assert_logits_close(got, want, 'G12 follower, bidirectional encoder, step 0 (float64 anchor)')
    ref_own = float(np.abs(g['logits_first'][np.isfinite(want)] - want[np.isfinite(want)]).max())
    print('[parity] the reference\'s own fp32 evaluation is %.3e from its float64 one' % ref_own)
    np.testing.assert_allclose(float(st.loss.detach()), g['loss'], rtol=1e-4)
    st.loss.backward()
    torch.cuda.synchronize()
    named = {k: p.grad for k, p in enc.named_parameters() if p.grad is not None}
    for k in ('embedding.weight', 'lstm.weight_ih_l0_reverse', 'lstm.weight_hh_l0_reverse', 'lstm.bias_hh_l0_reverse'):
        assert k in named, k
    assert float(named['embedding.weight'][0].abs().sum()) == 0.0              # padding row (model.py:55)
    check_grads(named, g, 'enc/')
    check_grads({k: p.grad for k, p in dec.named_parameters() if p.grad is not None}, g, 'dec/')


def test_bidirectional_inference_rollout_equals_the_module_composition():
    """FollowerEngine with a bidirectional encoder, eval mode: the engine's ctx / initial state are the module's."""
    from speaker_follower_amd import model, features, follower as fol
    d = synth.FULL
    enc = _encoder(9, glove=True).eval()
    _, dec_w = synth.follower_weights_peaky(515)
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    dec.cuda().eval()
    fb = synth.follower_batch(seed=41, batch=32, steps=5, n_viewpoints=64, min_len=3, max_len=30, stop_prob=0.05)
    store = features.FeatureStore(synth.feature_table(4, 64))
    eng = fol.FollowerEngine(enc, dec, store)
    batch = fol.DeviceFollowerBatch.from_synth(fb)
    with torch.no_grad():
        st = eng.rollout(batch, 5, 'argmax')
        ctx, h, c = enc(batch.seq, batch.lengths)
    torch.cuda.synchronize()
    assert torch.equal(st.ctx, ctx) and torch.equal(st.hs_all[0], h) and torch.equal(st.cs_all[0], c)
    assert torch.isfinite(st.logits[st.logits > -1e30]).all()


def test_captured_rollout_with_a_bidirectional_encoder_replays():
    """hipGraph capture of an inference rollout whose encoder is the two-direction composition (torch index ops and
    the C entries on the capture stream): a replay reproduces the eager rollout bit for bit."""
    from speaker_follower_amd import model, features, follower as fol
    d = synth.FULL
    enc = _encoder(9, glove=True).eval()
    _, dec_w = synth.follower_weights_peaky(515)
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    dec.cuda().eval()
    fb = synth.follower_batch(seed=43, batch=32, steps=5, n_viewpoints=64, min_len=3, max_len=30, stop_prob=0.05)
    store = features.FeatureStore(synth.feature_table(4, 64))
    eng = fol.FollowerEngine(enc, dec, store)
    batch = fol.DeviceFollowerBatch.from_synth(fb)
    with torch.no_grad():
        eager = eng.rollout(batch, 5, 'argmax')
    replay, st = eng.capture(batch, 5, 'argmax')
    replay()
    torch.cuda.synchronize()
    assert torch.equal(st.actions, eager.actions) and torch.equal(st.logits, eager.logits)
