"""GPU: the search procedures against outputs of the REFERENCE's own
beam_search / state_factored_search / speaker beam_search (tests/golden/g7_search.json, generated
by tests/golden/make_golden_search.py on the same seeded world)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import search_world as W          # noqa: E402

SCORE_TOL = 3e-4                  # a score is a sum of <= 12 log-probabilities, each within 1e-4 rel


@pytest.fixture(scope='module')
def golden():
    with open(os.path.join(HERE, 'golden', 'g7_search.json')) as f:
        return json.load(f)


@pytest.fixture(scope='module')
def world():
    from speaker_follower_amd import model, features, agents, synth
    env, table = W.build_world(dense=True)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights(W.FOLLOWER_SEED)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    agent = agents.Seq2SeqAgent(env, '/tmp/sf_search.json', enc, dec, episode_len=W.EPISODE_LEN)
    agent.store = features.FeatureStore(table)
    senc_w, sdec_w = synth.speaker_weights(W.SPEAKER_SEED)
    senc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    sdec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    senc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    sdec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    senc.cuda().eval()
    sdec.cuda().eval()
    speaker = agents.Seq2SeqSpeaker(env, '/tmp/sf_search_spk.json', senc, sdec, W.INSTRUCTION_LEN,
                                    max_episode_len=W.EPISODE_LEN)
    return env, agent, speaker


def check_candidates(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g['instr_id'] == w['instr_id']
        assert [int(a) for a in g['actions']] == w['actions']
        assert [p[0] for p in g['trajectory']] == w['viewpoints']
        assert abs(g['score'] - w['score']) <= SCORE_TOL * max(1.0, abs(w['score']))
        np.testing.assert_allclose(g['scores'], w['scores'], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('beam', [1, 3, 5])
def test_beam_search_matches_reference(world, golden, beam):
    env, agent, _ = world
    env.set_beam_size(beam)
    env.reset_epoch()
    got = []
    for _ in range(W.N_ITEMS // W.BATCH):
        trajs, completed, traversed = agent.beam_search(beam)
        assert traversed is None
        got += trajs
    want = golden['beam'][str(beam)]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        check_candidates(g, w)
    for tl in got:
        for c in tl:
            assert len(c['attentions']) == len(c['actions'])
            np.testing.assert_allclose([a.sum() for a in c['attentions']], 1.0, rtol=1e-5)


@pytest.mark.parametrize('sizes', [(3, 1), (4, 2)])
def test_state_factored_search_matches_reference(world, golden, sizes):
    env, agent, _ = world
    comp, succ = sizes
    env.set_beam_size(max(comp, succ))
    env.reset_epoch()
    got, trav = [], []
    for _ in range(W.N_ITEMS // W.BATCH):
        trajs, completed, traversed = agent.state_factored_search(comp, succ)
        got += trajs
        trav += [[s.world_state.viewpointId for s in tr] for tr in traversed]
    want = golden['state_factored']['%d_%d' % (comp, succ)]
    assert len(got) == len(want)
    for g, t, w in zip(got, trav, want):
        check_candidates(g, w['cands'])
        assert t == w['traversed']
        ends = [c['observations'][-1]['viewpoint'] for c in g]      # one candidate per end state
        keys = [(c['observations'][-1]['viewpoint'], c['observations'][-1]['heading']) for c in g]
        assert len(set(keys)) == len(keys), ends


def test_state_factored_search_batch64_k40_matches_reference():
    """BASELINE.json configs[4] at its size (rational_follower.py:42-47: state_factored_search(K = 40, 1)
    over a minibatch of 64) on the fixture graphs with "peaky" weights: the same completions in the
    same order, the same physical traversal, scores within 3e-4."""
    import gzip
    import time
    from speaker_follower_amd import model, features, agents, synth
    with gzip.open(os.path.join(HERE, 'golden', 'g7_search_b64_k40.json.gz'), 'rt') as f:
        gold = json.load(f)
    cfg = gold['config']
    assert cfg['batch'] == 64 and cfg['K'] == 40
    env, table = W.build_world(dense=True, n_items=W.BIG_ITEMS, batch=W.BIG_BATCH, item_seed=W.BIG_ITEM_SEED)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(W.BIG_FOLLOWER_SEED)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    agent = agents.Seq2SeqAgent(env, '/tmp/sf_search_big.json', enc, dec, episode_len=W.BIG_EPISODE_LEN)
    agent.store = features.FeatureStore(table)
    agent.tie_log = []                    # per expansion: score of the expanded state and of its frontier's runner-up
    env.set_beam_size(W.BIG_K)
    env.reset_epoch()
    torch.cuda.synchronize()
    t0 = time.time()
    trajs, completed, traversed = agent.state_factored_search(W.BIG_K, 1)
    torch.cuda.synchronize()
    dt = time.time() - t0
    want = gold['results']
    assert len(trajs) == len(want) == 64
    n_cands = 0
    tie_inst = np.concatenate([x[0] for x in agent.tie_log])
    tie_gap = np.concatenate([np.abs(x[1] - x[2]) for x in agent.tie_log])
    tie_ulp = np.concatenate([np.spacing(np.abs(x[1]).astype(np.float32)) for x in agent.tie_log])
    reordered = []
    for i, (g, tr, w) in enumerate(zip(trajs, traversed, want)):
        check_candidates(g, w['cands'])           # the completions, their order and their scores: always
        if [s.world_state.viewpointId for s in tr] != w['traversed']:
            # A best-first search expands the best frontier state; where two states score EQUAL to the last bit or two
            # of fp32 the order is decided by the summation order of the arithmetic, the reference's or any other
            # (3 933 expansions here, 1 of them with a gap below 1e-5: instruction 31, gap 0 at score -6.676 with the
            # gate product on the bf16 matrix cores, 1 ulp the other way in the reference; tools/search_tie_probe.py).
            # Such an instruction may walk its (identical) set of expansions in another order -- nothing else may.
            m = tie_inst == i
            assert m.any() and float((tie_gap[m] / tie_ulp[m]).min()) <= 2.0, (i, float((tie_gap[m] / tie_ulp[m]).min()))
            reordered.append(i)
        n_cands += len(g)
    assert len(reordered) <= 2, reordered
    print('traversal identical to the reference for %d of 64 instructions; fp32 ties (<= 2 ulp) re-ordered: %s'
          % (64 - len(reordered), reordered))
    assert n_cands == sum(len(w['cands']) for w in want) and n_cands > 64 * 20
    print('state-factored search, batch 64, K = 40: %.2f s here; the reference took %.1f s on %d CPU threads'
          % (dt, cfg['reference_cpu_seconds'], cfg['reference_threads']))
    # ---- the production path (no diagnostics hook): one hipGraph replay (search.GraphStep) + one native bookkeeping
    # call (sim/frontier_core.cpp) per iteration, against the same golden and against the numpy path above
    del agent.tie_log
    env.reset_epoch()
    torch.cuda.synchronize()
    t0 = time.time()
    trajs_n, completed_n, traversed_n = agent.state_factored_search(W.BIG_K, 1)
    torch.cuda.synchronize()
    dt_n = time.time() - t0
    assert agent._graph_steps, 'the production path did not run'
    differ = []
    for i, (g, tr, w, tr_np) in enumerate(zip(trajs_n, traversed_n, want, traversed)):
        check_candidates(g, w['cands'])
        for a, b in zip(g, trajs[i]):                       # and the numpy path's, to roundoff
            assert a['actions'] == b['actions'] and a['trajectory'] == b['trajectory']
            assert abs(a['score'] - b['score']) <= 2e-5
        if [s.world_state.viewpointId for s in tr] != w['traversed']:
            m = tie_inst == i
            assert m.any() and float((tie_gap[m] / tie_ulp[m]).min()) <= 2.0, i
            differ.append(i)
    assert len(differ) <= 2, differ
    print('production path (graph + native): %.3f s, traversal re-ordered at fp32 ties: %s' % (dt_n, differ))
    # ---- the STRICT summation order (runtime.strict_gate_product: the gate products on the fp32 MFMA, the kernel of
    # rounds 1-3) as its own captured step.  Same completions, order and scores; same traversal except at exact ties.
    # Which of the two orders reproduces the reference's walk at a 0-ulp tie is NOT a property of either kernel: round 3
    # (fp32) 64 / 64, round 4 (split) 63 / 64, round 5 split 64 / 64 and fp32 63 / 64 after an unrelated re-compilation
    # of the attention kernel -- the reference's own order at such a tie is its own roundoff.
    from speaker_follower_amd import runtime
    with runtime.strict_gate_product():
        agent.tie_log = []
        env.reset_epoch()
        _, _, trav_np = agent.state_factored_search(W.BIG_K, 1)
        s_inst = np.concatenate([x[0] for x in agent.tie_log])
        s_gap = np.concatenate([np.abs(x[1] - x[2]) for x in agent.tie_log])
        s_ulp = np.concatenate([np.spacing(np.abs(x[1]).astype(np.float32)) for x in agent.tie_log])
        del agent.tie_log
        env.reset_epoch()
        trajs_s, _, traversed_s = agent.state_factored_search(W.BIG_K, 1)
        torch.cuda.synchronize()
        assert any(k[-1] == 1 for k in agent._graph_steps), 'the strict step was not captured on its own'
    differ_s = []
    for i, (g, tr, w) in enumerate(zip(trajs_s, traversed_s, want)):
        check_candidates(g, w['cands'])
        if [s.world_state.viewpointId for s in tr] != w['traversed']:
            m = s_inst == i
            assert m.any() and float((s_gap[m] / s_ulp[m]).min()) <= 2.0, i
            differ_s.append(i)
    assert len(differ_s) <= 2, differ_s
    print('strict gate products: traversal identical to the reference for %d of 64 instructions, re-ordered at fp32 ties: %s'
          % (64 - len(differ_s), differ_s))


def test_beam_one_equals_greedy_rollout(world):
    """follower.py:150-156 (the reference's own commented sanity check): beam_search(1) reproduces
    the argmax rollout's trajectory and score."""
    env, agent, _ = world
    env.set_beam_size(1)
    env.reset_epoch()
    agent.feedback = 'argmax'
    with torch.no_grad():
        greedy = agent._rollout_with_loss()
    beams, _, _ = agent.beam_search(1, load_next_minibatch=False)
    assert len(beams) == len(greedy)
    for b, g in zip(beams, greedy):
        assert b[0]['instr_id'] == g['instr_id']
        assert b[0]['trajectory'] == g['trajectory']
        assert abs(b[0]['score'] - g['score']) < 2e-4 * max(1.0, abs(g['score']))


@pytest.mark.parametrize('beam', [1, 4])
def test_speaker_beam_search_matches_reference(world, golden, beam):
    env, _, speaker = world
    env.reset_epoch()
    path_obs, path_actions, _ = env.gold_obs_actions_and_instructions(W.EPISODE_LEN)
    outs = speaker.beam_search(beam, path_obs, path_actions)
    want = golden['speaker_beam'][str(beam)]
    assert len(outs) == len(want)
    for ol, wl in zip(outs, want):
        assert len(ol) == len(wl)
        for o, w in zip(ol, wl):
            assert o['instr_id'] == w['instr_id']
            assert o['word_indices'] == w['word_indices']
            assert abs(o['score'] - w['score']) <= SCORE_TOL * max(1.0, abs(w['score']))
            np.testing.assert_allclose(o['scores'], w['scores'], rtol=2e-4, atol=2e-4)
            assert len(o['attentions']) == len(o['word_indices'])


def test_rational_follower_reranks_candidates(world):
    from speaker_follower_amd import search
    env, agent, speaker = world
    res, counts = search.run_rational_follower(env, None, agent, speaker, beam_size=3,
                                               state_factored_search=True, physical_traversal=True)
    assert set(res) == {0.0, 0.95}
    for w in res:
        assert len(res[w]) == W.N_ITEMS
        assert sum(counts[w].values()) == W.N_ITEMS
        for instr_id, cand in res[w].items():
            assert cand['instr_id'] == instr_id and 'speaker_score' in cand and 'follower_score' in cand
    # weight 0 = follower score only: the first (best-scoring) candidate wins everywhere
    assert set(counts[0.0]) == {0}


def test_logprob_topk_kernel_against_torch():
    from speaker_follower_amd._lib import call
    from speaker_follower_amd.runtime import ptr, stream
    g = torch.Generator().manual_seed(3)
    for N, n, k in ((5, 7, 7), (33, 14, 5), (9, 991, 40), (4, 1024, 1)):
        ld = (n + 3) & ~3
        x = torch.randn(N, ld, generator=g).cuda()
        x[:, 1] = x[:, 0]                                          # ties: lower column first
        nv = torch.randint(1, n + 1, (N,), generator=g).to(torch.int32).cuda() if n < 100 else None
        ref = x[:, :n].clone()
        if nv is not None:
            ref[torch.arange(n, device='cuda')[None, :] >= nv[:, None]] = -float('inf')
        lp = torch.log_softmax(ref, 1)
        idx = torch.empty(N, k, dtype=torch.int32, device='cuda')
        logp = torch.empty(N, k, device='cuda')
        xin = x.clone()
        call('sf_logprob_topk', ptr(xin), ld, N, n, ptr(nv) if nv is not None else None, k, ptr(idx),
             ptr(logp), stream())
        order = torch.sort(ref, dim=1, descending=True, stable=True)[1][:, :k]
        assert torch.equal(idx.long(), order)
        torch.testing.assert_close(logp, lp.gather(1, order), rtol=1e-5, atol=1e-5)
        if nv is not None:
            assert torch.equal(xin[:, :n], ref)                    # masked in place


def test_logprob_rows_in_column_order():
    """sf_logprob_topk with idx = NULL: the whole row, log_softmax in column order, -inf beyond n_valid."""
    from speaker_follower_amd import _lib
    from speaker_follower_amd._lib import call
    from speaker_follower_amd.runtime import ptr, stream
    g = torch.Generator().manual_seed(5)
    for N, n in ((64, 14), (7, 5), (3, 991)):
        ld = (n + 3) & ~3
        x = torch.randn(N, ld, generator=g).cuda()
        nv = torch.randint(1, n + 1, (N,), generator=g).to(torch.int32).cuda()
        ref = x[:, :n].clone()
        ref[torch.arange(n, device='cuda')[None, :] >= nv[:, None]] = -float('inf')
        out = torch.full((N, n), 7.0, device='cuda')
        call('sf_logprob_topk', ptr(x.clone()), ld, N, n, ptr(nv), n, None, ptr(out), stream())
        want = torch.log_softmax(ref, 1)
        assert torch.equal(torch.isinf(out), torch.isinf(want))
        fin = ~torch.isinf(want)
        torch.testing.assert_close(out[fin], want[fin], rtol=1e-5, atol=1e-5)
        # the same numbers as the sorted form
        idx = torch.empty(N, n, dtype=torch.int32, device='cuda')
        logp = torch.empty(N, n, device='cuda')
        call('sf_logprob_topk', ptr(x.clone()), ld, N, n, ptr(nv), n, ptr(idx), ptr(logp), stream())
        ok = idx >= 0
        rows = torch.arange(N, device='cuda')[:, None].expand(N, n)
        assert torch.equal(out[rows[ok], idx.long()[ok]], logp[ok])
    rc = _lib.lib.sf_logprob_topk(ptr(x), ld, N, n, None, 3, None, ptr(out), stream())        # NULL idx needs k == n
    assert rc == _lib.SF_ERR_ARG


def test_scatter_rows_kernel():
    from speaker_follower_amd._lib import call
    from speaker_follower_amd.runtime import ptr, stream
    src = torch.randn(6, 512, device='cuda')
    idx = torch.tensor([40, 3, -1, 0, 49, -1], dtype=torch.int32, device='cuda')
    dst = torch.full((50, 516), 9.0, device='cuda')
    call('sf_scatter_rows', ptr(src), 512, ptr(idx), 6, 512, ptr(dst), 516, stream())
    want = torch.full((50, 516), 9.0, device='cuda')
    for i, d in enumerate(idx.tolist()):
        if d >= 0:
            want[d, :512] = src[i]
    assert torch.equal(dst, want)


def test_graph_step_equals_host_issued_steps_and_survives_pool_growth(world):
    """search.GraphStep (one hipGraph replay per iteration: nav look-ups on the device, instructions padded to the
    agent's maximum length) against search.FlatDecoder over host-packed inputs, for the same frontier: same
    log-probabilities to roundoff, same state rows; a pool that has to grow re-captures and keeps its rows."""
    from speaker_follower_amd import frontier, search, nav
    env, agent, _ = world
    env.reset_epoch()
    env_, space, fd, t, roots = frontier._setup(agent, True)
    inputs, _ = frontier._step_inputs(space, t, roots)
    base, logp = fd.step_logprobs(inputs)
    B = len(roots)
    env.reset_epoch()
    agent.__dict__.pop('_graph_steps', None)
    env2, space2, gs = frontier._setup_space(agent, True, graph_cap=B + 3)          # (capacity above the frontier)
    gs.pool_rows = 0                                                               # force _grow on the first run
    from speaker_follower_amd.sim import load_frontier
    h = space2.h
    core = load_frontier().StateFactored(10, 1, agent.episode_len, 4, 36, h['next_row'], h['cand_view'], h['a_num'],
                                         space2.base_row, space2.root_sid, space2.root_key)
    n = core.fill_inputs(gs.inputs, gs.n)
    assert n == B and (gs.inputs[7, n:] == -1).all()
    out = gs.run(n).copy()
    A = logp.shape[1]
    assert np.array_equal(np.isinf(out[:n, :A]), np.isinf(logp)) and np.isinf(out[:n, A:]).all()
    fin = ~np.isinf(logp)
    np.testing.assert_allclose(out[:n, :A][fin], logp[fin], rtol=0, atol=2e-5)
    torch.cuda.synchronize()
    np.testing.assert_allclose(gs.hpool[B:B + n].cpu().numpy(), fd.hpool.buf[base:base + n].cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(gs.hpool[:B].cpu().numpy(), fd.hpool.buf[:B].cpu().numpy(), atol=0)   # seeds survived _grow
    Tn = fd.apool.buf.shape[1]
    np.testing.assert_allclose(gs.apool[B:B + n, :Tn].cpu().numpy(), fd.apool.buf[base:base + n].cpu().numpy(), atol=2e-5)
    assert float(gs.apool[B:B + n, Tn:].abs().max()) == 0.0                        # padded instruction positions: weight 0
    agent.__dict__.pop('_graph_steps', None)


def test_gather_rows_kernel():
    from speaker_follower_amd._lib import call
    from speaker_follower_amd.runtime import ptr, stream
    src = torch.randn(50, 512, device='cuda')
    idx = torch.tensor([3, 3, 49, -1, 0, 17], dtype=torch.int32, device='cuda')
    dst = torch.empty(6, 512, device='cuda')
    call('sf_gather_rows', ptr(src), 512, ptr(idx), 6, 512, ptr(dst), 512, stream())
    want = src[idx.long().clamp(min=0)]
    want[3] = 0
    assert torch.equal(dst, want)


def test_speaker_rescoring_of_all_candidates_at_once_equals_small_batches(world):
    """rational_follower.py:67-69 scores ALL candidate routes of a minibatch as one batch (~2 500 rows at configs[4]:
    more than the persistent word loop's 128 rows and than sf_speaker_loss_finalize's 1 024 threads).  The one-batch
    result equals scoring the same routes in batches of 100 (the persistent launch), route for route."""
    from speaker_follower_amd import search
    env, agent, speaker = world
    env.reset_epoch()
    env.set_beam_size(40)
    with torch.no_grad():
        cands, _, _ = agent.state_factored_search(40, 1)
    flat = search.flatten(cands) * 9                                   # > 1 024 rows
    assert len(flat) > 1100
    args = ([c['observations'] for c in flat], [c['actions'] for c in flat], [c['instr_encoding'] for c in flat])
    with torch.no_grad():
        big, loss_big = speaker._score_obs_actions_and_instructions(*args, feedback='teacher')     # chunks of 128 rows
        type(speaker).SCORE_CHUNK, keep = 1 << 20, type(speaker).SCORE_CHUNK
        try:                                                                                          # ONE batch of all rows
            one, loss_one = speaker._score_obs_actions_and_instructions(*args, feedback='teacher')
        finally:
            type(speaker).SCORE_CHUNK = keep
        assert abs(float(loss_big) - float(loss_one)) <= 1e-4 * abs(float(loss_one))
        for a, b in zip(big, one):
            assert a['word_indices'] == b['word_indices'] and abs(a['score'] - b['score']) <= 2e-4 * max(1.0, abs(b['score']))
        small = []
        for lo in range(0, len(flat), 100):
            out, _ = speaker._score_obs_actions_and_instructions(*(a[lo:lo + 100] for a in args), feedback='teacher')
            small += out
    assert len(big) == len(small) == len(flat)
    for a, b, c in zip(big, small, flat):
        assert a['instr_id'] == b['instr_id'] == c['instr_id']
        assert a['word_indices'] == b['word_indices'] == ([int(x) for x in c['instr_encoding']] + [2])[:W.INSTRUCTION_LEN]
        assert abs(a['score'] - b['score']) <= 2e-4 * max(1.0, abs(b['score']))
        np.testing.assert_allclose(a['scores'], b['scores'], rtol=0, atol=2e-4)


def test_rational_follower_with_scores_issued_by_the_search_equals_the_plain_pipeline(world):
    """search.run_rational_follower hands the search's routes to the speaker in index form before their result
    dictionaries exist (Seq2SeqSpeaker.route_scores_hook): the scores collected by the later
    _score_obs_actions_and_instructions call equal those of the plain call over the dictionaries, route for route."""
    from speaker_follower_amd import search, agents, features
    _, dense_agent, dense_speaker = world
    env, table = W.build_world(dense=False)                    # index-form observations: the speaker scores on the device
    store = features.FeatureStore(table)
    agent = agents.Seq2SeqAgent(env, '/tmp/sf_search_ix.json', dense_agent.encoder, dense_agent.decoder,
                                episode_len=W.EPISODE_LEN)
    agent.store = store
    speaker = agents.Seq2SeqSpeaker(env, '/tmp/sf_search_ix_spk.json', dense_speaker.encoder, dense_speaker.decoder,
                                    W.INSTRUCTION_LEN, max_episode_len=W.EPISODE_LEN)
    speaker.store = store
    cls = type(speaker)
    keep = cls.SCORE_CHUNK
    try:
        cls.SCORE_CHUNK = 16                                   # (the fixture yields ~100 routes per minibatch)
        speaker.prefetch_hits = 0
        res_a, cnt_a = search.run_rational_follower(env, None, agent, speaker, beam_size=20, state_factored_search=True)
        assert speaker.prefetch_hits >= 1 and agent.candidates_hook is None
        cls.SCORE_CHUNK = 1 << 20                              # no chunks, no hook: one engine batch over the dictionaries
        speaker.prefetch_hits = 0
        res_b, cnt_b = search.run_rational_follower(env, None, agent, speaker, beam_size=20, state_factored_search=True)
        assert speaker.prefetch_hits == 0
    finally:
        cls.SCORE_CHUNK = keep
    assert cnt_a == cnt_b
    for w in res_a:
        assert set(res_a[w]) == set(res_b[w])
        for k in res_a[w]:
            a, b = res_a[w][k], res_b[w][k]
            assert a['actions'] == b['actions'] and a['trajectory'] == b['trajectory']
            assert abs(a['speaker_score'] - b['speaker_score']) <= 2e-4 * max(1.0, abs(b['speaker_score']))
            assert a['follower_score'] == b['follower_score']


def test_rational_speaker_pipeline_matches_reference():
    """rational_speaker.py:9-137 (`generate_and_score_candidates` + `predict_from_candidates`): golden G13 is the output of
    the reference's own module over this fixture world (tests/golden/make_golden_rational_speaker.py).  Per instruction
    the same candidate instructions in the same order (word ids identical), speaker and follower scores within 3e-4, the
    follower's teacher-forced actions identical; per speaker weight the same candidate is chosen wherever the reference's
    own margin between its two best candidates exceeds the score tolerance."""
    from speaker_follower_amd import model, features, agents, synth, search
    with open(os.path.join(HERE, 'golden', 'g13_rational_speaker.json')) as f:
        gold = json.load(f)
    cfg = gold['config']
    env, table = W.build_world(dense=True)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(cfg['follower_seed'])
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    follower = agents.Seq2SeqAgent(env, '/tmp/sf_rs.json', enc, dec, episode_len=cfg['episode_len'],
                                   max_instruction_length=cfg['instruction_len'])
    follower.store = features.FeatureStore(table)
    senc_w, sdec_w = synth.speaker_weights_peaky(cfg['speaker_seed'])
    senc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    sdec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    senc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    sdec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    senc.cuda().eval()
    sdec.cuda().eval()
    speaker = agents.Seq2SeqSpeaker(env, '/tmp/sf_rs_spk.json', senc, sdec, cfg['instruction_len'],
                                    max_episode_len=cfg['episode_len'])
    speaker.store = follower.store
    # (ONE epoch over a fresh environment, as the golden run: the environment reshuffles its items when it wraps around,
    # and the speaker encoder's padded path steps make a candidate list depend on its minibatch's longest path -- the
    # reference's own behaviour, model.py:445-451)
    by_id = search.generate_and_score_candidates(env, speaker, follower, cfg['n_candidates'])
    assert {str(k) for k in by_id} == set(gold['candidates'])
    worst_s = worst_f = 0.0
    for k, lst in by_id.items():
        want = gold['candidates'][str(k)]
        assert len(lst) == len(want)
        for c, w in zip(lst, want):
            assert [int(x) for x in c['word_indices']] == w['word_indices']
            assert [int(a) for a in c['actions']] == w['actions']
            worst_s = max(worst_s, abs(c['speaker_score'] - w['speaker_score']) / max(1.0, abs(w['speaker_score'])))
            worst_f = max(worst_f, abs(c['follower_score'] - w['follower_score']) / max(1.0, abs(w['follower_score'])))
    print('rational speaker: %d instructions, %d candidates; worst relative score difference speaker %.2e, follower %.2e'
          % (len(by_id), sum(len(v) for v in by_id.values()), worst_s, worst_f))
    assert worst_s <= SCORE_TOL and worst_f <= SCORE_TOL
    # the re-ranking: the reference's choice wherever its own top-two margin is not within the score tolerance
    ss = np.array([c['speaker_score'] for lst in gold['candidates'].values() for c in lst])
    fs = np.array([c['follower_score'] for lst in gold['candidates'].values() for c in lst])
    res = search.predict_from_candidates(by_id, [float(w) for w in np.arange(0, 21) / 20.0])
    agree = total = 0
    for w, chosen in res.items():
        sw, fw = w / ss.std(), (1 - w) / fs.std()
        for k, best in chosen.items():
            want = gold['candidates'][str(k)]
            mixed = sorted((c['speaker_score'] * sw + c['follower_score'] * fw for c in want), reverse=True)
            got = next(i for i, c in enumerate(by_id[k]) if c is best)
            total += 1
            if got == gold['chosen']['%.2f' % w][str(k)]:
                agree += 1
            else:
                assert mixed[0] - mixed[1] <= 1e-3 * max(1.0, abs(mixed[0])), (w, k, mixed[:2])
    print('re-ranking: %d of %d (weight, instruction) choices equal the reference\'s' % (agree, total))
    assert agree >= 0.97 * total
    # the whole pipeline in one call (rational_speaker.py:140-165) on a fresh world
    env2, _ = W.build_world(dense=True)
    scores, results = search.run_rational_speaker(env2, None, speaker, follower, cfg['n_candidates'])
    assert scores is None and len(results) == 21 and all(len(r) == W.N_ITEMS for r in results.values())
    for w in (0.0, 1.0):
        for k, best in results[w].items():
            key = 'follower_score' if w == 0.0 else 'speaker_score'
            assert best[key] == max(c[key] for c in by_id[k])
