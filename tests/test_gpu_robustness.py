"""GPU: the three silent-failure holes of round 3, closed.

* `sample` feedback of the speaker (speaker.py:170-174): the per-step glue kernel and the persistent word loop draw
  from softmax(logit) with the counter-based two-level sampler of csrc/sf_sampling.h -- checked against its float64
  mirror (oracle/rng.py) draw by draw and against softmax by chi-square; the sampled word really is the one fed back.
* a starved persistent launch (bounded wait given up -> NaN-poisoned outputs): the kernels raise a FAULT WORD in the
  workspace, the engines read it at their sync and re-issue the pass on the per-step kernels in the same process.
  Forced here with sf_debug_persist_timeout(0).
* the speaker's loss stops at the first step at which every row has produced EOS (speaker.py:192-197), on the
  index-form engine path as on the dense path.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import np_env, np_model, rng as orng                      # noqa: E402  (checker only)
from tests.tol import assert_logits_close                            # noqa: E402
from tests.test_gpu_persistent import speaker_setup, encoder, batch as instr_batch, run as run_encoder   # noqa: E402
from speaker_follower_amd import synth                                # noqa: E402

PAD, EOS = 0, 2


def _glue(logit, target, feedback, seed, stream, row0=0):
    from speaker_follower_amd import _lib
    from speaker_follower_amd.runtime import ptr, stream as cur
    B, V = logit.shape
    ldv = (V + 3) & ~3
    lg = torch.zeros(B, ldv, device='cuda')
    lg[:, :V] = logit
    w = torch.empty(B, dtype=torch.int64, device='cuda')
    score, nll, live = (torch.empty(B, device='cuda') for _ in range(3))
    ended = torch.zeros(B, dtype=torch.uint8, device='cuda')
    smp = C.byref(_lib.Sample(seed, stream, row0)) if feedback == 2 else None
    _lib.call('sf_speaker_glue_fwd', B, V, ldv, ptr(lg), ptr(target), feedback, PAD, EOS, ptr(ended), ptr(w), ptr(score),
              ptr(nll), ptr(live), smp, cur())
    torch.cuda.synchronize()
    return w.cpu().numpy(), score.cpu().numpy(), ended.cpu().numpy()


def test_speaker_glue_refuses_sample_without_a_generator():
    from speaker_follower_amd import _lib
    from speaker_follower_amd.runtime import ptr, stream as cur
    lg = torch.zeros(4, 992, device='cuda')
    t = torch.zeros(4, dtype=torch.int64, device='cuda')
    f = torch.zeros(4, device='cuda')
    e = torch.zeros(4, dtype=torch.uint8, device='cuda')
    rc = _lib.lib.sf_speaker_glue_fwd(4, 991, 992, ptr(lg), ptr(t), 2, PAD, EOS, ptr(e), ptr(t), ptr(f), ptr(f), ptr(f),
                                      None, cur())
    assert rc == _lib.SF_ERR_ARG


@pytest.mark.parametrize('vocab', [991, 1024, 37])
def test_speaker_glue_sample_equals_the_mirror_draw_by_draw(vocab):
    g = np.random.default_rng(vocab)
    B = 512
    logit = (g.standard_normal((B, vocab)) * g.choice([0.3, 1.5, 4.0], size=(B, 1))).astype(np.float32)
    target = torch.from_numpy(g.integers(0, vocab, B)).cuda()
    seed, stream, row0 = 0xC0FFEE, 17, 1000
    w, score, ended = _glue(torch.from_numpy(logit).cuda(), target, 2, seed, stream, row0)
    u1, u2 = orng.sample_uniforms(seed, stream, row0 + np.arange(B))
    want = [orng.speaker_sample(logit[b], u1[b], u2[b]) for b in range(B)]
    clear = np.array([m > 1e-5 for _, m in want])
    assert clear.mean() > 0.98
    ww = np.array([x for x, _ in want])
    assert np.array_equal(w[clear], ww[clear]), np.flatnonzero(w != ww)
    assert (w >= 0).all() and (w < vocab).all()
    logp = logit.astype(np.float64) - np.log(np.exp(logit.astype(np.float64) - logit.max(1, keepdims=True)).sum(1, keepdims=True)) \
        - logit.max(1, keepdims=True)
    want_score = np.where(w != PAD, logp[np.arange(B), w], 0.0)
    np.testing.assert_allclose(score, want_score, rtol=1e-4, atol=1e-4)            # speaker.py:179-180
    assert np.array_equal(ended, (w == EOS).astype(np.uint8))                     # :190-191


def _chi2(counts, p, n):
    from scipy import stats
    keep = p * n >= 5
    obs = np.append(counts[keep], counts[~keep].sum())
    exp = np.append(p[keep] * n, p[~keep].sum() * n)
    if exp[-1] < 5:
        obs[-2] += obs[-1]
        exp[-2] += exp[-1]
        obs, exp = obs[:-1], exp[:-1]
    chi2 = float(((obs - exp) ** 2 / exp).sum())
    return chi2, float(stats.chi2.sf(chi2, len(obs) - 1))


@pytest.mark.parametrize('temp', [0.5, 2.5])
def test_speaker_glue_sample_follows_softmax(temp):
    g = np.random.default_rng(5)
    vocab, N = 991, 40000
    row = (g.standard_normal(vocab) * temp).astype(np.float32)
    logit = torch.from_numpy(np.tile(row, (N, 1))).cuda()
    target = torch.zeros(N, dtype=torch.int64, device='cuda')
    w, _, _ = _glue(logit, target, 2, 99, 3)
    p = np.exp(row.astype(np.float64) - row.max())
    p /= p.sum()
    chi2, pval = _chi2(np.bincount(w, minlength=vocab).astype(np.float64), p, N)
    assert pval > 1e-4, (chi2, pval)


@pytest.mark.parametrize('B', [100, 37])
def test_persistent_speaker_sample_feedback(B):
    """The persistent word loop with feedback 'sample': every word equals the mirror's draw from the launch's own
    logits of that step (so the draw follows softmax), and the word fed back is the sampled one: teacher-forcing
    the per-step path on the sampled words reproduces the logits of every step."""
    from speaker_follower_amd import speaker
    enc, dec, store, batch = speaker_setup(B, peaky=True)
    S = 24
    eng = speaker.SpeakerEngine(enc, dec, store)
    eng.dropout_seed = 0x1234
    batch.row0 = 300
    with torch.no_grad():
        st = eng.score(batch, S, 'sample', train=False)
    torch.cuda.synchronize()
    assert st.persistent
    words = st.words.cpu().numpy()                                    # [S+1,B]
    logits = st.logits.cpu().numpy()                                  # [S,B,vocab]
    assert not np.isnan(logits).any()
    seed = (0x1234 ^ 0x3C6EF372) & 0xFFFFFFFF
    n_clear = n_all = 0
    for t in range(S):
        u1, u2 = orng.sample_uniforms(seed, st.site0 + t, 300 + np.arange(B))
        for b in range(B):
            w, margin = orng.speaker_sample(logits[t, b], u1[b], u2[b])
            n_all += 1
            if margin > 1e-5:
                n_clear += 1
                assert words[t + 1, b] == w, (t, b, words[t + 1, b], w, margin)
    assert n_clear > 0.98 * n_all
    assert len(np.unique(words[1:])) > 20                             # it is not the arg max in disguise
    # scores = log p(sampled word)
    lse = np.log(np.exp(logits.astype(np.float64) - logits.max(2, keepdims=True)).sum(2)) + logits.max(2)
    pick = np.take_along_axis(logits, words[1:, :, None], axis=2)[:, :, 0]
    want = np.where(words[1:] != PAD, pick - lse, 0.0)
    np.testing.assert_allclose(st.step_scores.cpu().numpy(), want, rtol=1e-4, atol=2e-4)
    # teacher-force the per-step path on the sampled words
    import copy
    b2 = copy.copy(batch)
    b2.instr_seq = torch.from_numpy(np.ascontiguousarray(words[1:].T)).cuda()
    ref = speaker.SpeakerEngine(enc, dec, store)
    ref.persistent = False
    with torch.no_grad():
        rt = ref.score(b2, S, 'teacher', train=False)
    assert not rt.persistent
    la, lb = rt.logits.cpu().numpy(), logits
    assert float(np.abs(la - lb).max()) <= 2e-4 * max(1.0, float(np.abs(la).max()))
    # sharding invariance: the second half of the rows alone draws the same words
    half = B // 2
    sb = synth.speaker_batch(seed=B + 1, batch=B, n_viewpoints=128, min_len=5, max_len=60)
    sub = type(sb)(**{k: (v[half:] if k in ('instr', 'path_len') else v[:, half:]) for k, v in sb.__dict__.items()})
    bh = speaker.DeviceSpeakerBatch.from_synth(sub, row0=300 + half)
    e2 = speaker.SpeakerEngine(enc, dec, store)
    e2.dropout_seed = 0x1234
    with torch.no_grad():
        sh = e2.score(bh, S, 'sample', train=False)
    wa, wb = sh.words.cpu().numpy(), words[:, half:]
    assert (wa == wb).mean() > 0.97          # (a draw within fp32 roundoff of a CDF boundary may flip and then diverge)


def test_per_step_speaker_sample_matches_the_persistent_loop():
    from speaker_follower_amd import speaker
    enc, dec, store, batch = speaker_setup(100, peaky=True)
    out = []
    for persistent in (True, False):
        eng = speaker.SpeakerEngine(enc, dec, store)
        eng.dropout_seed, eng.persistent = 77, persistent
        with torch.no_grad():
            out.append(eng.score(batch, 12, 'sample', train=False))
    a, b = (s.words.cpu().numpy() for s in out)
    assert out[0].persistent and not out[1].persistent
    assert (a == b).mean() > 0.97


@pytest.mark.parametrize('persistent', [True, False])
def test_speaker_argmax_loss_stops_where_every_row_has_ended(persistent):
    """speaker.py:192-197 (advisor, round 3): under argmax feedback the gold targets stay live behind the predicted
    EOS; the reference leaves its loop once every row has ended, so those steps must not be added."""
    from speaker_follower_amd import model, features, speaker
    d = synth.FULL
    senc_w, sdec_w = synth.speaker_weights_peaky(404)
    # make EOS the arg max early for every row: a large bias on the EOS column from the start
    sdec_w = dict(sdec_w)
    bias = sdec_w['decoder2action.bias'].copy()
    bias[EOS] += 11.0                  # (the oracle then leaves its loop after 11 of 40 steps, rows ending at 0 .. 10)
    sdec_w['decoder2action.bias'] = bias
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    B, S = 24, 40
    sb = synth.speaker_batch(seed=3, batch=B, n_viewpoints=64, min_len=30, max_len=60)
    table = synth.feature_table(8, 64)
    store = features.FeatureStore(table)
    eng = speaker.SpeakerEngine(enc, dec, store)
    eng.persistent = persistent
    with torch.no_grad():
        st = eng.score(speaker.DeviceSpeakerBatch.from_synth(sb), S, 'argmax', train=False)
    loc = np_env.static_loc_embeddings()
    acts, feats, pmask = np_env.dense_speaker_inputs(sb, table, loc)
    seq, _, _ = np_env.batch_instructions_from_encoded(sb.instr, 80)
    ref = np_model.speaker_score(senc_w, sdec_w, acts, feats, pmask, seq, S, 'argmax')
    n = len(ref['logits'])
    assert 1 <= n < S - 5, n                                           # the reference really left its loop early
    words = st.words.cpu().numpy()
    assert np.array_equal(words[1:n + 1], ref['words'])
    np.testing.assert_allclose(float(st.loss), float(ref['loss']), rtol=1e-4)
    # ... and summing every step would have been a different number
    full = float((st.sum_cnt[:, 0] / st.sum_cnt[:, 1].clamp(min=1)).sum())
    assert full > float(ref['loss']) * 1.2
    assert float(st.gscale[n:].abs().max()) == 0.0 and float(st.gscale[:n].min()) > 0.0


# ---------------------------------------------------------------------------------------- fault word
@pytest.fixture
def forced_timeout():
    from speaker_follower_amd import _lib, runtime
    runtime.take_fault(torch.device('cuda', 0))
    _lib.lib.sf_debug_persist_timeout(0)
    yield
    _lib.lib.sf_debug_persist_timeout(-1)
    torch.cuda.synchronize()
    runtime.take_fault(torch.device('cuda', 0))


def _follower(seed=11):
    from speaker_follower_amd import model
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return enc.cuda(), dec.cuda(), enc_w, dec_w


def test_a_starved_encoder_launch_raises_the_fault_word(forced_timeout):
    from speaker_follower_amd import runtime
    enc = encoder()
    seq, mask, lens = instr_batch(7, 100, 10, 60)
    got = run_encoder(enc, seq, lens, persistent=True)
    torch.cuda.synchronize()
    assert torch.isnan(got['ctx']).any()                               # poisoned, as before ...
    bits = runtime.take_fault(seq.device)
    assert bits & runtime.FAULT_ENC_FWD                                # ... and now the host can see it
    assert runtime.take_fault(seq.device) == 0                         # read-and-clear


def test_follower_run_falls_back_to_the_per_step_kernels(forced_timeout):
    """FollowerEngine.run under a forced timeout: the rollout is re-issued on the per-step encoder kernels in the same
    process and equals the oracle (actions bit-exact, logits 1e-4)."""
    from speaker_follower_amd import features, follower as fol
    enc, dec, enc_w, dec_w = _follower()
    enc.eval()
    dec.eval()
    B, S, NVP = 40, 6, 64
    fb = synth.follower_batch(seed=5, batch=B, steps=S, n_viewpoints=NVP, min_len=8, max_len=40)
    table = synth.feature_table(3, NVP)
    eng = fol.FollowerEngine(enc, dec, features.FeatureStore(table))
    batch = fol.DeviceFollowerBatch.from_synth(fb)
    with torch.no_grad():
        raw = eng.rollout(batch, S, 'argmax', train=False)
        torch.cuda.synchronize()
        assert torch.isnan(raw.loss)                                   # what used to be returned with rc 0
        st = eng.run(batch, S, 'argmax', train=False)
    assert eng.fallbacks == 1 and getattr(enc, 'persistent', True)     # the switch was restored
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, 80, reverse=True)
    loc = np_env.static_loc_embeddings()
    ref = np_model.follower_rollout(enc_w, dec_w, seq, lens, mask, S,
                                    lambda t: np_env.dense_follower_step(table, loc, fb, t),
                                    fb.target, 'argmax', synth.FULL.feat, early_exit=False)
    assert np.array_equal(st.actions.cpu().numpy(), ref['actions'])
    lg = st.logits.cpu().numpy()
    for t in range(S):
        a = ref['logits'][t].shape[1]
        assert_logits_close(lg[t][:, :a], ref['logits'][t], 'fallback rollout, step %d' % t)
    np.testing.assert_allclose(float(st.loss), float(ref['loss']), rtol=1e-4)


def test_training_iteration_survives_a_starved_backward(forced_timeout):
    """run(backward=True): forward AND backward persistent launches time out; the iteration is replayed with the
    per-step kernels and leaves the gradients of an undisturbed iteration (same dropout sites)."""
    from speaker_follower_amd import _lib, features, follower as fol
    B, S, NVP = 32, 5, 64
    fb = synth.follower_batch(seed=8, batch=B, steps=S, n_viewpoints=NVP, min_len=8, max_len=40)
    table = synth.feature_table(3, NVP)
    grads = []
    for faulty in (False, True):
        enc, dec, _, _ = _follower()
        enc.train()
        dec.train()
        eng = fol.FollowerEngine(enc, dec, features.FeatureStore(table))
        eng.dropout_seed = 4242
        _lib.lib.sf_debug_persist_timeout(0 if faulty else -1)
        st = eng.run(fol.DeviceFollowerBatch.from_synth(fb), S, 'teacher', train=True, backward=True)
        torch.cuda.synchronize()
        assert eng.fallbacks == (1 if faulty else 0)
        assert torch.isfinite(st.loss)
        grads.append({k: p.grad.clone() for k, p in list(enc.named_parameters()) + list(dec.named_parameters())
                      if p.grad is not None})
    assert set(grads[0]) == set(grads[1]) and len(grads[0]) > 10
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert torch.isfinite(b).all(), k
        s = float(a.abs().max())
        # (persistent bf16-split products vs per-step fp32 MFMA kernels; biases whose true gradient is zero hold 1e-8 roundoff)
        assert float((a - b).abs().max()) <= 1e-4 * max(s, 1e-3), k


def test_speaker_run_falls_back_to_the_per_step_kernels(forced_timeout):
    from speaker_follower_amd import speaker, _lib
    enc, dec, store, batch = speaker_setup(50, peaky=True)
    eng = speaker.SpeakerEngine(enc, dec, store)
    with torch.no_grad():
        raw = eng.score(batch, 20, 'argmax', train=False)
        torch.cuda.synchronize()
        assert raw.persistent and torch.isnan(raw.step_scores).any()
        st = eng.run(batch, 20, 'argmax', train=False)
    assert eng.fallbacks == 1 and not st.persistent
    _lib.lib.sf_debug_persist_timeout(-1)
    with torch.no_grad():
        good = speaker.SpeakerEngine(enc, dec, store).score(batch, 20, 'argmax', train=False)
    assert good.persistent
    assert torch.equal(good.words, st.words)
    np.testing.assert_allclose(st.step_scores.cpu().numpy(), good.step_scores.cpu().numpy(), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize('B', [7, 100, 1024, 1025, 2600])
def test_speaker_loss_finalize_for_any_batch_size(B):
    """sf_speaker_loss_finalize directly (speaker.py:192-197): the step means are added up to and including the first
    step at which EVERY row has produced EOS -- also for more rows than the kernel has threads (the pragmatic
    re-ranking scores all ~2 500 candidate routes of a minibatch as one batch, rational_follower.py:67-69)."""
    from speaker_follower_amd import _lib
    from speaker_follower_amd.runtime import ptr, stream as cur
    T = 37
    g = np.random.default_rng(B)
    words = g.integers(3, 900, size=(T + 1, B)).astype(np.int64)
    first = g.integers(2, 20, size=B)
    first[g.integers(B)] = 29                                   # the slowest row
    for b in range(B):
        words[1 + first[b], b] = EOS                            # (row 0 of `words` = the start tokens)
        if b % 3 == 0:
            words[1 + min(first[b] + 4, T - 1), b] = EOS        # a later EOS does not matter
    sum_cnt = np.stack((g.random(T) * 50 + 1, g.integers(1, B + 1, size=T)), 1).astype(np.float32)
    sum_cnt[5, 1] = 0.0                                         # a step without live rows adds nothing
    loss = torch.zeros(1, device='cuda')
    gscale = torch.full((T,), -1.0, device='cuda')
    d_sc, d_w = torch.tensor(sum_cnt).cuda(), torch.tensor(words).cuda()
    _lib.call('sf_speaker_loss_finalize', ptr(d_sc), ptr(d_w), EOS, T, B, ptr(loss), ptr(gscale), cur())
    torch.cuda.synchronize()
    last = int(first.max())
    assert last == 29
    want = np.float32(0)
    want_g = np.zeros(T, np.float32)
    for t in range(last + 1):
        if sum_cnt[t, 1] > 0:
            want = np.float32(want + np.float32(sum_cnt[t, 0] / sum_cnt[t, 1]))
            want_g[t] = np.float32(1.0) / sum_cnt[t, 1]
    assert float(loss) == float(want)
    assert np.array_equal(gscale.cpu().numpy(), want_g)
    # no row ever ends: all T steps count
    words[words == EOS] = 5
    d_w = torch.tensor(words).cuda()
    _lib.call('sf_speaker_loss_finalize', ptr(d_sc), ptr(d_w), EOS, T, B, ptr(loss), ptr(gscale), cur())
    torch.cuda.synchronize()
    assert int((gscale.cpu().numpy() > 0).sum()) == T - 1       # (every step but the one without live rows)


def _index_world():
    """Index-form env + agents over the search fixture graphs (the speaker scores on the device)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import search_world as W
    from speaker_follower_amd import model, features, agents
    env, table = W.build_world(dense=False)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights(W.FOLLOWER_SEED)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    senc_w, sdec_w = synth.speaker_weights(W.SPEAKER_SEED)
    senc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    sdec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    senc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    sdec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    for m in (enc, dec, senc, sdec):
        m.cuda().eval()
    store = features.FeatureStore(table)
    agent = agents.Seq2SeqAgent(env, '/tmp/sf_rob_agent.json', enc, dec, episode_len=W.EPISODE_LEN)
    agent.store = store
    speaker = agents.Seq2SeqSpeaker(env, '/tmp/sf_rob_speaker.json', senc, sdec, W.INSTRUCTION_LEN,
                                    max_episode_len=W.EPISODE_LEN)
    speaker.store = store
    env.set_beam_size(20)
    return env, agent, speaker


def _routes(agent, env):
    from speaker_follower_amd import search
    env.reset_epoch()
    with torch.no_grad():
        cands, _, _ = agent.state_factored_search(20, 1)
    flat = search.flatten(cands)
    return [c['observations'] for c in flat], [c['actions'] for c in flat], [c['instr_encoding'] for c in flat]


def test_scores_issued_for_other_routes_are_not_collected():
    """Seq2SeqSpeaker.route_scores_hook issued a sweep for the search's routes; a scoring call over DIFFERENT observation
    lists (here: the same routes in another order, as new list objects) must not collect it -- it drops the pending
    sweep and scores what it was given."""
    env, agent, speaker = _index_world()
    cls = type(speaker)
    keep = cls.SCORE_CHUNK
    try:
        cls.SCORE_CHUNK = 16
        agent.candidates_hook = speaker.route_scores_hook('teacher')
        obs, acts, instr = _routes(agent, env)
        assert getattr(speaker, '_pending_scores', None) is not None and len(obs) > 40
        order = list(range(len(obs)))[::-1]
        speaker.prefetch_hits = 0
        with torch.no_grad():
            out, _ = speaker._score_obs_actions_and_instructions([list(obs[i]) for i in order], [acts[i] for i in order],
                                                                  [instr[i] for i in order], feedback='teacher')
        assert speaker.prefetch_hits == 0 and getattr(speaker, '_pending_scores', None) is None
        agent.candidates_hook = None
        with torch.no_grad():
            ref, _ = speaker._score_obs_actions_and_instructions(obs, acts, instr, feedback='teacher')
        for k, i in enumerate(order):
            assert out[k]['instr_id'] == ref[i]['instr_id'] and out[k]['word_indices'] == ref[i]['word_indices']
            assert abs(out[k]['score'] - ref[i]['score']) <= 2e-4 * max(1.0, abs(ref[i]['score']))
    finally:
        cls.SCORE_CHUNK = keep


def test_chunked_route_scoring_survives_starved_persistent_launches(forced_timeout):
    """Every persistent word-loop launch of a chunked scoring sweep gives up its first wait (forced): the fault word is
    read once at the end of the sweep and the whole sweep re-issued on the per-step kernels -- same scores as a healthy
    run, `fallbacks` counted."""
    from speaker_follower_amd import _lib
    env, agent, speaker = _index_world()
    _lib.lib.sf_debug_persist_timeout(-1)                     # (a healthy search and a healthy reference first)
    obs, acts, instr = _routes(agent, env)
    cls = type(speaker)
    keep = cls.SCORE_CHUNK
    try:
        cls.SCORE_CHUNK = 16
        with torch.no_grad():
            ref, _ = speaker._score_obs_actions_and_instructions(obs, acts, instr, feedback='teacher')
        assert speaker._engine.fallbacks == 0
        _lib.lib.sf_debug_persist_timeout(0)
        with torch.no_grad():
            out, _ = speaker._score_obs_actions_and_instructions(obs, acts, instr, feedback='teacher')
        assert speaker._engine.fallbacks == 1
    finally:
        cls.SCORE_CHUNK = keep
    for a, b in zip(out, ref):
        assert a['word_indices'] == b['word_indices'] and np.isfinite(a['score'])
        assert abs(a['score'] - b['score']) <= 2e-4 * max(1.0, abs(b['score']))


def test_route_scoring_as_replayed_graphs_equals_the_launch_by_launch_sweep(forced_timeout):
    """Seq2SeqSpeaker.score_graphs: the full chunks of a large teacher-forced scoring sweep as replayed graphs on two
    streams (speaker.SpeakerSweep with scores), the remainder launch by launch -- every route's words and scores and the
    loss of the launch-by-launch sweep, on first use (captures) and on replay; a starved launch re-issues everything on
    the per-step kernels."""
    from speaker_follower_amd import _lib
    env, agent, speaker = _index_world()
    _lib.lib.sf_debug_persist_timeout(-1)
    obs, acts, instr = _routes(agent, env)
    cls = type(speaker)
    keep = cls.SCORE_CHUNK
    try:
        cls.SCORE_CHUNK = 16
        assert len(obs) // 16 >= 4 and len(obs) % 16 != 0                 # full chunks + a remainder
        with torch.no_grad():
            ref, loss_ref = speaker._score_obs_actions_and_instructions(obs, acts, instr, feedback='teacher')
            speaker.score_graphs = True
            first, loss_first = speaker._score_obs_actions_and_instructions(obs, acts, instr, feedback='teacher')
            again, loss_again = speaker._score_obs_actions_and_instructions(obs, acts, instr, feedback='teacher')
            assert len(speaker._score_sweeps) == 1 and speaker._engine.fallbacks == 0
            _lib.lib.sf_debug_persist_timeout(0)                           # every persistent launch gives up its first wait
            starved, _ = speaker._score_obs_actions_and_instructions(obs, acts, instr, feedback='teacher')
            assert speaker._engine.fallbacks == 1
    finally:
        cls.SCORE_CHUNK = keep
        speaker.score_graphs = False
        _lib.lib.sf_debug_persist_timeout(-1)
    assert abs(float(loss_first) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref)) and float(loss_again) == float(loss_first)
    for a, b, c, d in zip(ref, first, again, starved):
        assert a['instr_id'] == b['instr_id'] == c['instr_id'] == d['instr_id']
        assert a['word_indices'] == b['word_indices'] == c['word_indices'] == d['word_indices']
        assert a['score'] == b['score'] == c['score'] and a['scores'] == b['scores']
        assert abs(d['score'] - a['score']) <= 2e-4 * max(1.0, abs(a['score']))


def test_a_capture_never_creates_its_workspace_inside_the_graph():
    """runtime.workspace under stream capture: a workspace first touched inside a capture would be zero-filled (64 MB) by
    EVERY replay -- it rode in every captured rollout until round 5.  Creating one under capture raises; the engines
    create their capture stream's workspace first (runtime.ensure_workspace) and their captures go through."""
    from speaker_follower_amd import runtime
    dev = torch.device('cuda', 0)
    # a stream that has no workspace yet (torch recycles stream handles: one handed out before may come back)
    keep = []
    for _ in range(64):
        fresh = torch.cuda.Stream()
        keep.append(fresh)
        if (0, fresh.cuda_stream) not in runtime._workspaces:
            break
    else:
        pytest.skip('every stream handle torch hands out already has a workspace')
    g = torch.cuda.CUDAGraph()
    raised = False
    with torch.cuda.stream(fresh):
        with torch.cuda.graph(g, stream=fresh):
            try:
                runtime.workspace(dev)
            except RuntimeError as exc:
                raised = 'ensure_workspace' in str(exc)
    assert raised
    other = torch.cuda.Stream()
    runtime.ensure_workspace(other, dev)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(other):
        with torch.cuda.graph(g2, stream=other):
            ws = runtime.workspace(dev)                     # exists: nothing is allocated or filled in the graph
    assert ws.numel() == runtime.lib.sf_workspace_bytes()


def test_the_collector_is_off_while_a_stream_captures():
    """runtime.graph_capture: a cyclic collection inside a capture may finalise an older graph / stream / event, which HIP
    refuses while a stream captures -- raised in a destructor, that aborts the process (seen under pytest once the
    agents' loops moved the collector's thresholds).  Every capture of the package goes through this context."""
    import gc
    from speaker_follower_amd import runtime
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    x = torch.zeros(8, device='cuda')
    torch.cuda.synchronize()
    assert gc.isenabled()
    with runtime.graph_capture(g, s):
        assert not gc.isenabled()
        x += 1
    assert gc.isenabled()
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert x.tolist() == [2.0] * 8
    import inspect
    from speaker_follower_amd import follower, search, speaker
    for mod in (follower, search, speaker, runtime):
        src = inspect.getsource(mod)
        assert 'with torch.cuda.graph(' not in src.replace('with torch.cuda.graph(graph, stream=stream_)', ''), mod.__name__


def test_a_training_pass_releases_its_tape_without_the_collector():
    """follower._RolloutLossFn / speaker._SpeakerLossFn: state -> loss -> grad_fn -> ctx -> state is a reference cycle; the
    backward drops ctx.state, so a finished iteration's tape goes when its last reference goes -- not whenever the cyclic
    collector next runs (a training loop showed a 4 GB sawtooth).  A second backward over the same pass is refused."""
    import gc
    from speaker_follower_amd import synth, model, features, follower
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights(9)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().train()
    dec.cuda().train()
    fb = synth.follower_batch(seed=3, batch=32, steps=6, n_viewpoints=40, min_len=4, max_len=33, a_max=9)
    store = features.FeatureStore(synth.feature_table(3, 40))
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    eng = follower.FollowerEngine(enc, dec, store)

    def iteration():
        st = eng.rollout(batch, 6, 'teacher', train=True)
        st.loss.backward()
        return st
    st = iteration()                                     # (gradients, workspaces, caches exist from here on)
    with pytest.raises(RuntimeError, match='second backward'):
        st.loss.backward()
    del st
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        st = iteration()
        held = torch.cuda.memory_allocated() - base
        del st
        left = torch.cuda.memory_allocated() - base
    finally:
        gc.enable()
    print('[tape] one iteration holds %.1f MB, %.1f MB after its state is dropped (collector off)' % (held / 2 ** 20, left / 2 ** 20))
    assert held > (1 << 20) and left <= held // 20
