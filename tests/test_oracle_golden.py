"""The oracle (numpy forward, torch-CPU fwd+bwd) against the golden vectors produced
by the reference modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import np_env, np_model, torch_ref
from speaker_follower_amd import synth

TOL = dict(rtol=2e-5, atol=2e-5)


def _sub(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------- G6
def test_loc_table_matches_reference(golden):
    g = golden('g6_env')
    np.testing.assert_array_equal(np_env.static_loc_embeddings(), g['loc_table'])


def test_action_embedding_matches_reference(golden):
    g = golden('g6_env')
    emb = np_env.action_embedding(g['act_feats'], g['act_cand_view'], g['act_cand_heading'],
                                  g['act_cand_elevation'])
    np.testing.assert_array_equal(emb, g['act_embedding'])


def test_batch_instructions_matches_reference(golden):
    g = golden('g6_env')
    sizes = g['instr_sizes']
    toks = np.split(g['instr_tokens'], np.cumsum(sizes)[:-1])
    for tag, rev in (('fwd', False), ('rev', True)):
        seq, mask, lens = np_env.batch_instructions_from_encoded(toks, 80, reverse=rev)
        np.testing.assert_array_equal(seq, g['instr_seq_' + tag])
        np.testing.assert_array_equal(mask, g['instr_mask_' + tag])
        np.testing.assert_array_equal(lens, g['instr_len_' + tag])
    seq, mask, lens, perm = np_env.batch_instructions_from_encoded(toks, 80, reverse=True, sort=True)
    np.testing.assert_array_equal(lens, g['instr_len_sorted'])
    np.testing.assert_array_equal(seq, g['instr_seq_sorted'])


# ----------------------------------------------------------------------------- G1
def test_lstm_cell_small(golden):
    g = _sub(golden('g1_modules_small'), 'lstm/')
    h1, c1 = np_model.lstm_cell(g['x'], g['h'], g['c'], g['weight_ih'], g['weight_hh'],
                                g['bias_ih'], g['bias_hh'])
    np.testing.assert_allclose(h1, g['h1'], **TOL)
    np.testing.assert_allclose(c1, g['c1'], **TOL)
    w = {k: torch.tensor(g[k], requires_grad=True) for k in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')}
    x, h, c = (torch.tensor(g[k], requires_grad=True) for k in ('x', 'h', 'c'))
    th1, tc1 = torch_ref.lstm_cell(x, h, c, w['weight_ih'], w['weight_hh'], w['bias_ih'], w['bias_hh'])
    ((th1 * torch.tensor(g['gh'])).sum() + (tc1 * torch.tensor(g['gc'])).sum()).backward()
    np.testing.assert_allclose(x.grad, g['dx'], **TOL)
    np.testing.assert_allclose(h.grad, g['dh'], **TOL)
    np.testing.assert_allclose(c.grad, g['dc'], **TOL)
    for k, p in w.items():
        np.testing.assert_allclose(p.grad, g['d_' + k], **TOL)


def test_soft_dot_attention_small(golden):
    g = _sub(golden('g1_modules_small'), 'sda/')
    ht, alpha = np_model.soft_dot_attention(g['h'], g['ctx'], g['mask'], g['linear_in.weight'],
                                            g['linear_out.weight'])
    np.testing.assert_allclose(ht, g['h_tilde'], **TOL)
    np.testing.assert_allclose(alpha, g['alpha'], **TOL)
    assert np.all(alpha[g['mask']] == 0)
    w_in = torch.tensor(g['linear_in.weight'], requires_grad=True)
    w_out = torch.tensor(g['linear_out.weight'], requires_grad=True)
    h = torch.tensor(g['h'], requires_grad=True)
    ctx = torch.tensor(g['ctx'], requires_grad=True)
    tht, _ = torch_ref.soft_dot_attention(h, ctx, torch.tensor(g['mask']), w_in, w_out)
    (tht * torch.tensor(g['go'])).sum().backward()
    np.testing.assert_allclose(h.grad, g['dh'], **TOL)
    np.testing.assert_allclose(ctx.grad, g['dctx'], **TOL)
    np.testing.assert_allclose(w_in.grad, g['d_linear_in.weight'], **TOL)
    np.testing.assert_allclose(w_out.grad, g['d_linear_out.weight'], **TOL)


def test_visual_attention_small(golden):
    g = _sub(golden('g1_modules_small'), 'vsda/')
    names = ('linear_in_h.weight', 'linear_in_h.bias', 'linear_in_v.weight', 'linear_in_v.bias')
    out, alpha = np_model.visual_soft_dot_attention(g['h'], g['X'], *(g[n] for n in names))
    np.testing.assert_allclose(out, g['out'], **TOL)
    np.testing.assert_allclose(alpha, g['alpha'], **TOL)
    w = [torch.tensor(g[n], requires_grad=True) for n in names]
    h = torch.tensor(g['h'], requires_grad=True)
    X = torch.tensor(g['X'], requires_grad=True)
    tout, _ = torch_ref.visual_soft_dot_attention(h, X, *w)
    (tout * torch.tensor(g['go'])).sum().backward()
    np.testing.assert_allclose(h.grad, g['dh'], **TOL)
    np.testing.assert_allclose(X.grad, g['dX'], **TOL)
    for n, p in zip(names, w):
        np.testing.assert_allclose(p.grad, g['d_' + n], **TOL)


def test_eltwise_prod_scoring_small(golden):
    g = _sub(golden('g1_modules_small'), 'eps/')
    names = ('linear_in_h.weight', 'linear_in_h.bias', 'linear_in_a.weight', 'linear_in_a.bias',
             'linear_out.weight', 'linear_out.bias')
    logit = np_model.eltwise_prod_scoring(g['h'], g['U'], *(g[n] for n in names))
    np.testing.assert_allclose(logit, g['logit'], **TOL)
    w = [torch.tensor(g[n], requires_grad=True) for n in names]
    h = torch.tensor(g['h'], requires_grad=True)
    U = torch.tensor(g['U'], requires_grad=True)
    (torch_ref.eltwise_prod_scoring(h, U, *w) * torch.tensor(g['go'])).sum().backward()
    np.testing.assert_allclose(h.grad, g['dh'], **TOL)
    np.testing.assert_allclose(U.grad, g['dU'], **TOL)
    for n, p in zip(names, w):
        np.testing.assert_allclose(p.grad, g['d_' + n], **TOL)


# ----------------------------------------------------------------------------- G2/G3
@pytest.fixture(scope='module')
def follower_setup():
    enc_w, dec_w = synth.follower_weights(101)
    fb8 = synth.follower_batch(seed=7, batch=8, steps=10, n_viewpoints=64, min_len=3,
                               max_len=19, a_max=8)
    table64 = synth.feature_table(7, 64)
    return enc_w, dec_w, fb8, table64, np_env.static_loc_embeddings()


def test_encoder_full_dims(golden, follower_setup):
    enc_w, dec_w, fb8, table64, loc = follower_setup
    g = golden('g3_encoder')
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb8.instr, 80, reverse=True)
    np.testing.assert_array_equal(lens, g['lengths'])
    ctx, h0, c0 = np_model.encoder_lstm(enc_w, seq, lens)
    np.testing.assert_allclose(ctx, g['ctx'], **TOL)
    np.testing.assert_allclose(h0, g['decoder_init'], **TOL)
    np.testing.assert_allclose(c0, g['c_t'], **TOL)


def test_decoder_step_full_dims(golden, follower_setup):
    enc_w, dec_w, fb8, table64, loc = follower_setup
    g3, g = golden('g3_encoder'), golden('g2_decoder_step')
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb8.instr, 80, reverse=True)
    X, all_u, is_valid = np_env.dense_follower_step(table64, loc, fb8, 0)
    h1, c1, alpha, logit, alpha_v = np_model.attn_decoder_step(
        dec_w, g['u_prev'], all_u, X, g3['decoder_init'], g3['c_t'], g3['ctx'], mask)
    np.testing.assert_allclose(h1, g['h1'], **TOL)
    np.testing.assert_allclose(c1, g['c1'], **TOL)
    np.testing.assert_allclose(alpha, g['alpha'], **TOL)
    np.testing.assert_allclose(alpha_v, g['alpha_v'], **TOL)
    np.testing.assert_allclose(logit, g['logit'], rtol=1e-4, atol=1e-4)


# ----------------------------------------------------------------------------- G4
def _np_rollout(enc_w, dec_w, fb, table, loc, steps, feedback):
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, 80, reverse=True)
    return np_model.follower_rollout(
        enc_w, dec_w, seq, lens, mask, steps,
        lambda t: np_env.dense_follower_step(table, loc, fb, t), fb.target, feedback, 2176)


def _check_rollout(res, g, logit_tol=1e-4):
    n = int(g['n_steps'])
    assert len(res['logits']) == n
    np.testing.assert_array_equal(res['actions'], g['actions'])      # bit-exact argmax
    for t in range(n):
        a = res['logits'][t].shape[1]
        ref = g['logits'][t][:, :a]
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(res['logits'][t]), fin)
        np.testing.assert_allclose(res['logits'][t][fin], ref[fin], rtol=logit_tol, atol=logit_tol)
    np.testing.assert_allclose(res['loss'], g['loss'], rtol=1e-4)
    np.testing.assert_allclose(res['scores'], g['scores'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(res['h'], g['h'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('feedback', ['teacher', 'argmax'])
def test_rollout_b8(golden, follower_setup, feedback):
    enc_w, dec_w, fb8, table64, loc = follower_setup
    res = _np_rollout(enc_w, dec_w, fb8, table64, loc, 10, feedback)
    _check_rollout(res, golden('g4_rollout_b8_' + feedback))


def _check_grads(named_grads, g, prefix, rtol=2e-3):
    seen = 0
    for name, grad in named_grads.items():
        key = prefix + 'gnorm/' + name
        if key not in g:
            continue
        seen += 1
        flat = grad.detach().numpy().ravel()
        norm = np.sqrt(np.sum(flat.astype(np.float64) ** 2))
        if g[key] < 1e-6:
            # visual linear_in_v.bias, decoder2action.linear_in_a.bias / linear_out.bias shift
            # every score of a row equally; softmax / CE are shift-invariant, so the true
            # gradient is 0 and the reference holds pure roundoff there.
            assert norm < 1e-5, name
            continue
        np.testing.assert_allclose(norm, g[key], rtol=rtol, err_msg=name)
        vals = flat[g[prefix + 'gidx/' + name]]
        np.testing.assert_allclose(vals, g[prefix + 'gval/' + name], rtol=rtol,
                                   atol=rtol * g[key] / np.sqrt(flat.size) + 1e-7, err_msg=name)
    assert seen > 0


def test_rollout_b8_teacher_gradients(golden, follower_setup):
    """torch_ref (the backward oracle) against the reference's autograd gradients."""
    enc_w, dec_w, fb8, table64, loc = follower_setup
    g = golden('g4_rollout_b8_teacher')
    enc = torch_ref.to_torch(enc_w, True, frozen=('embedding.weight',))
    dec = torch_ref.to_torch(dec_w, True)
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb8.instr, 80, reverse=True)
    res = torch_ref.follower_rollout(
        enc, dec, torch.tensor(seq), lens, torch.tensor(mask), 10,
        lambda t: np_env.dense_follower_step(table64, loc, fb8, t),
        torch.tensor(fb8.target), 'teacher', 2176)
    np.testing.assert_allclose(res['loss'].item(), g['loss'], rtol=1e-5)
    res['loss'].backward()
    _check_grads({k: v.grad for k, v in enc.items() if v.grad is not None}, g, 'enc/')
    _check_grads({k: v.grad for k, v in dec.items() if v.grad is not None}, g, 'dec/')


def test_rollout_b100_argmax_headline_shape(golden):
    """Full headline shape (B=100, 20 steps): oracle actions bit-exact, logits 1e-4."""
    enc_w, dec_w = synth.follower_weights(101)
    fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=256)
    table = synth.feature_table(0, 256)
    res = _np_rollout(enc_w, dec_w, fb, table, np_env.static_loc_embeddings(), 20, 'argmax')
    _check_rollout(res, golden('g4_rollout_b100_argmax'))


# ----------------------------------------------------------------------------- G5
@pytest.fixture(scope='module')
def speaker_setup():
    senc_w, sdec_w = synth.speaker_weights(202)
    sb = synth.speaker_batch(seed=9, batch=6, n_viewpoints=64, min_len=3, max_len=25)
    table64 = synth.feature_table(7, 64)
    loc = np_env.static_loc_embeddings()
    acts, feats, path_mask = np_env.dense_speaker_inputs(sb, table64, loc)
    instr_seq, _, _ = np_env.batch_instructions_from_encoded(sb.instr, 80)
    return senc_w, sdec_w, acts, feats, path_mask, instr_seq


@pytest.mark.parametrize('feedback,steps', [('teacher', 80), ('argmax', 30)])
def test_speaker_scoring(golden, speaker_setup, feedback, steps):
    senc_w, sdec_w, acts, feats, path_mask, instr_seq = speaker_setup
    g = golden('g5_speaker_b6_' + feedback)
    res = np_model.speaker_score(senc_w, sdec_w, acts, feats, path_mask, instr_seq, steps, feedback)
    n = int(g['n_steps'])
    assert len(res['logits']) == n
    np.testing.assert_array_equal(res['words'], g['words'])
    np.testing.assert_allclose(res['ctx'], g['ctx'], **TOL)
    np.testing.assert_allclose(np.stack(res['logits'][:3]), g['logits_first'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(res['logits'][-1], g['logit_last'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(res['loss'], g['loss'], rtol=1e-4)
    np.testing.assert_allclose(res['scores'], g['scores'], rtol=1e-4, atol=1e-4)


def test_speaker_teacher_gradients(golden, speaker_setup):
    senc_w, sdec_w, acts, feats, path_mask, instr_seq = speaker_setup
    g = golden('g5_speaker_b6_teacher')
    enc = torch_ref.to_torch(senc_w, True)
    dec = torch_ref.to_torch(sdec_w, True, frozen=('embedding.weight',))
    res = torch_ref.speaker_score(enc, dec, [torch.tensor(a) for a in acts],
                                  [torch.tensor(f) for f in feats], torch.tensor(path_mask),
                                  torch.tensor(instr_seq), 80, 'teacher')
    np.testing.assert_allclose(res['loss'].item(), g['loss'], rtol=1e-5)
    res['loss'].backward()
    _check_grads({k: v.grad for k, v in enc.items() if v.grad is not None}, g, 'enc/')
    _check_grads({k: v.grad for k, v in dec.items() if v.grad is not None}, g, 'dec/')


# ----------------------------------------------------------------------------- G8 / G9 (round 2)
# The hard cases of tests/golden/make_golden_hard.py: "peaky" weights (O(1) logits), all 20 steps
# of the headline batch, train mode with explicit dropout masks, speaker at B = 100.
def _scaled_close(got, want, scale_tol=1e-4):
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    scale = float(np.abs(want[fin]).max())
    assert float(np.abs(got[fin] - want[fin]).max()) <= scale_tol * scale, scale


def test_peaky_rollout_b100_all_twenty_steps(golden):
    g = golden('g8_follower_peaky_b100_argmax')
    enc_w, dec_w = synth.follower_weights_peaky(int(g['weight_seed']))
    fb = synth.follower_batch(seed=int(g['batch_seed']), batch=100, steps=20, n_viewpoints=256)
    table = synth.feature_table(int(g['table_seed']), 256)
    res = _np_rollout(enc_w, dec_w, fb, table, np_env.static_loc_embeddings(), 20, 'argmax')
    assert int(g['n_steps']) == 20 and len(res['logits']) == 20
    np.testing.assert_array_equal(res['actions'], g['actions'])
    for t in range(20):
        a = res['logits'][t].shape[1]
        _scaled_close(res['logits'][t], g['logits'][t][:, :a])
    np.testing.assert_allclose(res['loss'], g['loss'], rtol=1e-4)
    np.testing.assert_allclose(res['h'], g['h'], rtol=1e-4, atol=1e-4)


def test_peaky_train_mode_b100_loss_and_gradients(golden):
    """torch_ref with the SAME counter-based dropout masks (oracle/rng.py) against the reference
    modules run with those masks in place of nn.Dropout."""
    from oracle import rng as orng
    g = golden('g8_follower_peaky_b100_train')
    enc_w, dec_w = synth.follower_weights_peaky(int(g['weight_seed']))
    fb = synth.follower_batch(seed=int(g['batch_seed']), batch=100, steps=20, n_viewpoints=256,
                              stop_prob=1.0 / 40.0)
    table = synth.feature_table(int(g['table_seed']), 256)
    loc = np_env.static_loc_embeddings()
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, 80, reverse=True)
    B, T, H, F = 100, max(lens), 512, 2176
    seed, site0, rows = int(g['dropout_seed']), int(g['site0']), np.arange(100)

    def masks(t):
        if t == 'ctx':
            return torch.tensor(orng.dropout_mask(seed ^ 0x5BD1E995, site0, rows, T * H, 0.5).reshape(B, T, H))
        return (torch.tensor(orng.dropout_mask(seed, 2 * (site0 + t), rows, 2 * F, 0.5)),
                torch.tensor(orng.dropout_mask(seed, 2 * (site0 + t) + 1, rows, H, 0.5)))

    enc = torch_ref.to_torch(enc_w, True, frozen=('embedding.weight',))
    dec = torch_ref.to_torch(dec_w, True)
    res = torch_ref.follower_rollout(enc, dec, torch.tensor(seq), lens, torch.tensor(mask), 20,
                                     lambda t: np_env.dense_follower_step(table, loc, fb, t),
                                     torch.tensor(fb.target), 'teacher', F, drop_masks=masks)
    np.testing.assert_allclose(res['loss'].item(), g['loss'], rtol=1e-5)
    res['loss'].backward()
    gmax = max(float(v) for k, v in g.items() if 'gnorm/' in k)
    for prefix, named in (('enc/', enc), ('dec/', dec)):
        seen = 0
        for name, p in named.items():
            key = prefix + 'gnorm/' + name
            if p.grad is None or key not in g:
                continue
            seen += 1
            flat = p.grad.numpy().ravel()
            norm = np.sqrt(np.sum(flat.astype(np.float64) ** 2))
            if g[key] < 1e-6 * gmax:
                assert norm < 1e-5 * gmax, name
                continue
            np.testing.assert_allclose(norm, g[key], rtol=2e-3, err_msg=name)
            np.testing.assert_allclose(flat[g[prefix + 'gidx/' + name]], g[prefix + 'gval/' + name], rtol=2e-3,
                                       atol=2e-3 * g[key] / np.sqrt(flat.size) + 1e-7, err_msg=name)
        assert seen > 0


@pytest.mark.parametrize('feedback', ['teacher', 'argmax'])
def test_speaker_b100(golden, feedback):
    g = golden('g9_speaker_b100_' + feedback)
    senc_w, sdec_w = synth.speaker_weights_peaky(int(g['weight_seed']))
    sb = synth.speaker_batch(seed=int(g['batch_seed']), batch=100, n_viewpoints=256, min_len=10, max_len=79)
    table = synth.feature_table(int(g['table_seed']), 256)
    acts, feats, path_mask = np_env.dense_speaker_inputs(sb, table, np_env.static_loc_embeddings())
    instr_seq, _, _ = np_env.batch_instructions_from_encoded(sb.instr, 80)
    n = int(g['n_steps'])
    res = np_model.speaker_score(senc_w, sdec_w, acts, feats, path_mask, instr_seq, n, feedback)
    assert len(res['logits']) == n
    np.testing.assert_array_equal(res['words'], g['words'])
    _scaled_close(res['logits'][0], g['logits_first'][0])
    _scaled_close(res['logits'][-1], g['logit_last'])
    np.testing.assert_allclose(res['loss'], g['loss'], rtol=1e-4)
    np.testing.assert_allclose(res['scores'], g['scores'], rtol=1e-4, atol=2e-3)


# ----------------------------------------------------------------------------- G10: SR on real R2R items
def test_eval_restatement_reproduces_the_reference_evaluation():
    """oracle score_results (eval.py:56-139 restated over this repo's NavGraph distances) applied to the
    REFERENCE agent's trajectories gives the reference Evaluation's per-item errors and its summary
    (success rate, oracle rate, navigation error) on 156 real R2R instructions."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import r2r_world
    items, gold = r2r_world.load()
    _, _, graphs = r2r_world.build_env(items, gold['config'], dense=False)
    results = {k: dict(trajectory=[(vp, 0.0, 0.0) for vp in v['viewpoints']]) for k, v in gold['items'].items()}
    summary, per_item = np_env.score_results(r2r_world.gt_of(items), graphs, results)
    assert len(per_item) == 156 == gold['config']['n_items']
    for k, v in gold['items'].items():
        assert per_item[k]['success'] == v['success'] and per_item[k]['oracle_success'] == v['oracle_success']
        assert per_item[k]['steps'] == v['steps']
        np.testing.assert_allclose(per_item[k]['nav_error'], v['nav_error'], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(per_item[k]['length'], v['length'], rtol=1e-9, atol=1e-9)
    for k in ('success_rate', 'oracle_rate', 'nav_error', 'oracle_error', 'steps', 'lengths'):
        np.testing.assert_allclose(summary[k], gold['summary'][k], rtol=1e-9)
    assert summary['success_rate'] == 10 / 156


def test_speaker_decoder_input_att_feed_oracle_vs_reference(golden):
    """G14: SpeakerDecoderLSTM(use_input_att_feed=True) of the reference (model.py:500-513), three chained word steps;
    oracle/np_model.speaker_decoder_step_att_feed restates it."""
    from speaker_follower_amd import synth
    g = golden('g14_speaker_att_feed')
    dec = synth.speaker_decoder_att_feed_weights(int(g['weight_seed']))
    h, c = g['h0'], g['c0']
    for t in range(3):
        h, c, alpha, logit = np_model.speaker_decoder_step_att_feed(dec, g['words'][t], h, c, g['ctx'], g['mask'])
        np.testing.assert_allclose(h, g['h1_%d' % t], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(c, g['c1_%d' % t], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(alpha, g['alpha_%d' % t], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(logit, g['logit_%d' % t], rtol=1e-4, atol=1e-4)
