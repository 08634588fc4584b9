"""CPU: host logic of the search procedures (no compute calls) and the beamed env interface."""
import json
import os
import sys
from collections import namedtuple

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import search_world as W          # noqa: E402

WS = namedtuple('WS', 'scanId viewpointId heading elevation')


def _chain(search, vps, scores):
    s = None
    for i, (v, sc) in enumerate(zip(vps, scores)):
        s = search.InferenceState(s, WS('s', v, 0.0, 0.0), dict(viewpoint=v, heading=0.0, elevation=0.0),
                                  None, -1 if i == 0 else 1, None, i, sc, i, i, None if i == 0 else i)
    return s


def test_backchain_and_common_viewpoint_path():
    from speaker_follower_amd import search
    a = _chain(search, ['r', 'x', 'a1', 'a2'], [0.0, -1.0, -1.5, -2.25])
    states, obs, actions, scores, att = search.backchain_inference_states(a)
    assert [s.viewpointId for s in states] == ['r', 'x', 'a1', 'a2']
    assert actions == [1, 1, 1] and att == [1, 2, 3]
    np.testing.assert_allclose(scores, [-1.0, -0.5, -0.75])
    # b shares the prefix r, x: walking a -> b goes back to x and forward to b
    b_parent = a.prev_inference_state.prev_inference_state                      # the 'x' state
    b = search.InferenceState(b_parent, WS('s', 'b1', 0, 0), dict(viewpoint='b1'), None, 2, None, 2,
                              -3.0, 9, 9, 9)
    path = search.least_common_viewpoint_path(a, b)
    assert [s.world_state.viewpointId for s in path] == ['a2', 'a1', 'x', 'b1']


def test_scores_accumulate_in_float32_like_the_reference():
    from speaker_follower_amd import search
    s = 0.0
    for v in (np.float32(-0.1), np.float32(-0.2), np.float32(-0.7)):
        s = search._add_f32(s, v)
    want = np.float32(np.float32(np.float32(0) + np.float32(-0.1)) + np.float32(-0.2)) + np.float32(-0.7)
    assert s == float(want)


def test_rational_mix_standardises_and_picks():
    from speaker_follower_amd import search
    cands = {'a': [dict(follower_score=-1.0, speaker_score=-30.0), dict(follower_score=-2.0, speaker_score=-10.0)],
             'b': [dict(follower_score=-0.5, speaker_score=-20.0), dict(follower_score=-3.0, speaker_score=-21.0)]}
    res0, cnt0 = search.rational_mix(cands, 0.0)
    assert res0['a'] is cands['a'][0] and res0['b'] is cands['b'][0] and cnt0[0] == 2
    res1, cnt1 = search.rational_mix(cands, 0.95)
    assert res1['a'] is cands['a'][1] and res1['b'] is cands['b'][0]
    assert cnt1[1] == 1 and cnt1[0] == 1


def test_env_beamed_interface_matches_flat_interface():
    env, _ = W.build_world(dense=False)
    env.set_beam_size(3)
    assert env.beam_size == 3
    ws_b = env.reset(sort=True, beamed=True)
    ws = env.reset(sort=True, load_next_minibatch=False)
    assert [[w] for w in ws] == ws_b
    obs_b = env.observe(ws_b, beamed=True)
    obs = env.observe(ws)
    assert [o[0]['viewpoint'] for o in obs_b] == [o['viewpoint'] for o in obs]
    # two hypotheses per instance: stop, and the first neighbour
    beams = [[w, w] for w in ws]
    last = [[o, o] for o in obs]
    nxt = env.step(beams, [[0, 1]] * len(ws), last, beamed=True)
    for (stay, move), o, w in zip(nxt, obs, ws):
        assert stay == w
        assert move.viewpointId == o['adj_loc_list'][1]['nextViewpointId']
    flat = env.step(ws, [1] * len(ws), obs)
    assert flat == [m for _, m in nxt]


def test_golden_search_file_is_consistent():
    """Every reference candidate's score is the float32 sum of its per-step scores and ends with the
    stop action or at the episode limit (follower.py:663-666)."""
    with open(os.path.join(HERE, 'golden', 'g7_search.json')) as f:
        g = json.load(f)
    T = g['config']['episode_len']
    n = 0
    for beam, res in g['beam'].items():
        assert len(res) == g['config']['n_items']
        for cands in res:
            assert 1 <= len(cands) <= int(beam)
            assert all(a['score'] >= b['score'] for a, b in zip(cands, cands[1:]))
            for c in cands:
                assert c['actions'][-1] == 0 or len(c['actions']) == T
                assert len(c['viewpoints']) == len(c['actions']) + 1
                np.testing.assert_allclose(sum(c['scores']), c['score'], rtol=1e-5, atol=1e-5)
                n += 1
    assert n > 50


# ---------------------------------------------------------------- frontier.py: array form of the same logic
class _FakeSpace:
    """Just enough of frontier.StateSpace for Hyp views: a world state per (inst, sid)."""

    def world_state(self, inst, sid, start_pose):
        return WS('s', 'vp%d' % (int(sid) // 36), float(int(sid) % 36), 0.0)

    def observation(self, inst, sid, start_pose):
        return dict(viewpoint='vp%d' % (int(sid) // 36), heading=float(int(sid) % 36), elevation=0.0)


def _random_tree(rng, n_nodes, n_vp):
    from speaker_follower_amd import frontier
    t = frontier.Hypotheses(cap=4)                       # small capacity: exercises the growth path
    t.append(np.zeros(1, np.float32), np.ones(1, bool), parent=-1, inst=0, sid=int(rng.integers(n_vp)) * 36,
             key=0, action=-1, count=0, pool=0)
    while t.n < n_nodes:
        par = int(rng.integers(t.n))
        if t.count[par] >= 6:
            continue
        t.append(np.array([t.score[par] - rng.random()], np.float32), np.zeros(1, bool), parent=par, inst=0,
                 sid=int(rng.integers(n_vp)) * 36 + int(rng.integers(36)), key=0, action=1 + int(rng.integers(3)),
                 count=t.count[par] + 1, pool=t.n)
    return t


def test_array_physical_walk_equals_the_pairwise_lineage_walk():
    """frontier.physical_walks (all instances, all pairs at once, numpy) against chaining
    search.least_common_viewpoint_path over hypothesis views, on random hypothesis trees."""
    from speaker_follower_amd import frontier, search
    rng = np.random.default_rng(4)
    space = _FakeSpace()
    for trial in range(20):
        t = _random_tree(rng, 60, n_vp=5)
        root_vp = t.sid[0] // 36
        # a visiting order that always has a shared viewpoint: every node's lineage contains the root
        visits = [[0] + [int(x) for x in rng.integers(1, t.n, size=12)] for _ in range(3)]
        got = frontier.physical_walks(t, visits, depth=6)
        for seq, walk in zip(visits, got):
            want = [frontier.Hyp(t, space, seq[0])]
            for a, b in zip(seq, seq[1:]):
                path = search.least_common_viewpoint_path(frontier.Hyp(t, space, a), frontier.Hyp(t, space, b))
                # (two hypotheses standing on the same viewpoint: no movement, the path is [a] alone)
                assert path[0].node == a and path[-1].world_state.viewpointId == frontier.Hyp(t, space, b).world_state.viewpointId
                want += path[1:]
            assert walk == [h.node for h in want], (trial, seq)
        assert root_vp == t.sid[0] // 36


def test_hypothesis_views_backchain_like_the_namedtuple_states():
    from speaker_follower_amd import frontier, search
    rng = np.random.default_rng(9)
    t = _random_tree(rng, 40, n_vp=4)
    space = _FakeSpace()
    leaf = int(np.argmax(t.count[:t.n]))
    states, obs, actions, scores, att = search.backchain_inference_states(frontier.Hyp(t, space, leaf))
    lin = t.lineage(leaf)[::-1]
    assert len(states) == len(lin) and actions == [int(t.action[n]) for n in lin[1:]]
    assert att == [int(t.pool[n]) for n in lin[1:]]
    np.testing.assert_allclose(scores, np.diff(t.score[lin].astype(np.float64)))
    L, ln = frontier._lineage_matrix(t, [leaf], 6)
    assert ln[0] == len(lin) and L[0, :ln[0]].tolist() == lin[::-1]


def test_ragged_helpers():
    from speaker_follower_amd import frontier
    off, owner = frontier._ragged_arange(np.array([2, 0, 3]))
    assert off.tolist() == [0, 1, 0, 1, 2] and owner.tolist() == [0, 0, 2, 2, 2]
    g = np.array([0, 0, 0, 2, 2, 5])
    assert frontier._first_k_per_group(g, 2).tolist() == [True, True, False, True, True, True]
    assert frontier._first_k_per_group(np.zeros(0, int), 3).tolist() == []


# ---- the native bookkeeping of the state-factored search against its numpy restatement ----------------------
class _FakeStore:
    n, F, V = 0, 2176, 36

    def __init__(self):
        import torch
        self.device = torch.device('cpu')


class _FakeDecoder:
    """Stands where search.FlatDecoder stands: log-probabilities that are a deterministic function of the state and
    the hypothesis' history, QUANTISED to 1/4 so that exactly equal scores are common (the tie rules are what the
    two implementations could disagree on)."""

    def __init__(self, *a):
        self.base = 0

    def seed(self, h, c):
        self.base = h.shape[0]

    def step_logprobs(self, inp):
        n, a_max = len(inp['vp']), int(inp['a_num'].max())
        out = np.full((n, a_max), -np.inf, np.float32)
        for i in range(n):
            r = np.random.default_rng(int(inp['vp'][i]) * 36 + int(inp['view'][i]) + 977 * int(inp['crow'][i]))
            k = int(inp['a_num'][i])
            out[i, :k] = -np.round(r.random(k) * 12) / 4 - 0.25
        base = self.base
        self.base += n
        return base, out

    def attention_rows(self, rows):
        return [np.zeros(4, np.float32) for _ in rows]


def _fake_agent(monkeypatch, episode_len=6):
    import torch
    from speaker_follower_amd import search
    env, _ = W.build_world(dense=False, n_items=24, batch=24, item_seed=7)
    monkeypatch.setattr(search, 'FlatDecoder', _FakeDecoder)
    monkeypatch.setattr(search, '_require_store', lambda a: None)
    monkeypatch.setattr(search, '_encode_items', lambda agent, items: (
        torch.zeros(len(items), 4, 8), torch.zeros(len(items), 4), torch.zeros(len(items), 8), torch.zeros(len(items), 8)))

    class Agent:
        pass
    agent = Agent()
    agent.env, agent.store, agent.episode_len, agent.decoder = env, _FakeStore(), episode_len, None
    env.set_beam_size(8)
    return agent


def _search_record(out):
    trajs, completed, traversed = out
    return ([[(c['instr_id'], c['trajectory'], c['actions'], c['score'], c['scores']) for c in cands] for cands in trajs],
            [lst.nodes for lst in completed], [lst.nodes for lst in traversed])


def test_native_state_factored_bookkeeping_equals_the_numpy_restatement(monkeypatch):
    """sim/frontier_core.cpp against frontier._state_factored_search_numpy: the same hypotheses node for node
    (ids, scores, completion order, physical traversal) for several completion / successor sizes and state keys,
    with many exactly tied scores."""
    from speaker_follower_amd import frontier
    agent = _fake_agent(monkeypatch)
    n_ties = 0
    for k, s, key_fields in ((10, 1, 4), (40, 1, 4), (6, 3, 4), (12, 2, 3), (5, 1, 2)):
        recs = []
        for backend in ('numpy', 'native'):
            agent.search_backend = backend
            agent.env.reset_epoch()
            out = frontier.state_factored_search(agent, k, s, first_n_ws_key=key_fields)
            recs.append(_search_record(out))
            if backend == 'numpy':
                t = out[1][0].t
                sc = np.sort(t.score[:t.n])
                n_ties += int((np.diff(sc) == 0).sum())
        assert recs[0] == recs[1], (k, s, key_fields)
        assert all(len(c) >= 1 for c in recs[0][1])
    assert n_ties > 100                                   # the case really exercises the tie rules
    agent.search_backend = 'nonsense'
    agent.tie_log = []                                    # the diagnostics hook runs the numpy path
    agent.env.reset_epoch()
    frontier.state_factored_search(agent, 5, 1)
    assert agent.tie_log


def test_rational_speaker_reranking_matches_reference():
    """rational_speaker.py:109-137 (`predict_from_candidates`) is host arithmetic: fed the REFERENCE's candidates and
    scores (golden G13, tests/golden/make_golden_rational_speaker.py) it must choose the reference's candidate for all 21
    speaker weights x 16 instructions."""
    import json
    import os
    from speaker_follower_amd import search
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g13_rational_speaker.json')) as f:
        gold = json.load(f)
    by_id = {k: [dict(c) for c in lst] for k, lst in gold['candidates'].items()}
    weights = [float(w) for w in np.arange(0, 21) / 20.0]
    res = search.predict_from_candidates(by_id, weights)
    assert len(res) == 21
    for w in weights:
        for k, best in res[w].items():
            got = next(i for i, c in enumerate(by_id[k]) if c is best)
            assert got == gold['chosen']['%.2f' % w][k], (w, k)


def test_result_dictionaries_build_their_expensive_fields_on_demand():
    """frontier.Candidate: a search result (follower.py:694-716) whose 'observations' / 'trajectory' are made when read;
    every dict operation the reference's callers use (rational_follower.py:67-96: item reads, `del`, `in`, json.dump)
    must behave like the plain dictionary."""
    import copy
    import json
    import pickle
    from speaker_follower_amd import frontier

    built = []

    class Routes:
        lens = [3]
        inst = np.zeros((1, 4), np.int64)

        class space:
            items = [{'instr_id': '7_0'}]

        def trajectory(self, i):
            built.append('trajectory')
            return [('a', 0.0, 0.0), ('b', 0.5, 0.0), ('c', 1.0, 0.0)]

        def observations(self, i):
            built.append('observations')
            return [{'viewpoint': v} for v in 'abc']

    plain = {'instr_id': '7_0', 'score': 1.0, 'trajectory': Routes().trajectory(0), 'observations': Routes().observations(0)}
    built.clear()
    c = frontier.Candidate(Routes(), 0, {'instr_id': '7_0', 'score': 1.0})
    assert 'observations' in c and 'trajectory' in c and c['score'] == 1.0 and not built
    obs = c['observations']
    assert obs is c['observations'] and len(obs) == 3 and obs.instr_id == '7_0' and not built     # still nothing built
    assert obs[-1]['viewpoint'] == 'c' and [o['viewpoint'] for o in obs[:-1]] == ['a', 'b'] and 'trajectory' not in built
    assert c == plain and plain == dict(c) and sorted(c) == sorted(plain) and len(c) == 4
    assert isinstance(pickle.loads(pickle.dumps(c))['observations'], list) and copy.deepcopy(c) == plain
    del c['observations']
    assert 'observations' not in c and c.get('observations') is None
    with pytest.raises(KeyError):
        c['observations']
    assert json.loads(json.dumps(c))['trajectory'][1] == ['b', 0.5, 0.0]
    # deleting a field nobody read builds nothing
    built.clear()
    d = frontier.Candidate(Routes(), 0, {'instr_id': '7_0'})
    del d['observations']
    d['trajectory'] = ['walked']
    assert dict(d) == {'instr_id': '7_0', 'trajectory': ['walked']} and not built
