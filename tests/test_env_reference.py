"""CPU: this repo's environment layer against the REFERENCE's own env.py.

tests/golden/g15_env_reference.json.gz holds what the reference's `R2RBatch` / `EnvBatch` / `_get_panorama_states` /
`_navigate_to_location` / `_shortest_path_action` / `gold_obs_actions_and_instructions` (tasks/R2R/env.py:126-224,
723-761, 763-854; networkx all-pairs Dijkstra as the planner) produced on the real R2R_sub_val_seen split
(tests/golden/make_golden_env.py).  Checked against it:
  * `env.R2RIndexEnv`            minibatch order, panorama sweep (order of adj_loc_list, representative views, relative
                                 angles), teacher, env.step, the gold routes;
  * `sim/sweep_py.cpp` via `nav.NavTable`   the batched native sweep and the hop tables (host copies of what the
                                 device holds; the device kernels are checked in tests/test_gpu_nav_reference.py)."""
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import r2r_val_seen as VS                                                   # noqa: E402


@pytest.fixture(scope='module')
def world():
    from speaker_follower_amd import nav
    items, _ = VS.load_items()
    gold = VS.load_env_golden()
    env, row_of, n = VS.build_env(items, batch_size=gold['config']['batch'])
    table = nav.NavTable(env, types.SimpleNamespace(device=torch.device('cpu')))
    return items, gold, env, table


def test_the_split_is_the_one_the_reference_loaded(world):
    items, gold, env, table = world
    assert len(items) == gold['config']['n_items'] == 782 and len({it['scan'] for it in items}) == 51
    assert len({it['path_id'] for it in items}) == 260
    assert table.n_rows == sum(int(np.sum(g.included)) for g in env.graphs.values())


def test_minibatch_order_equals_r2rbatch(world):
    """random.seed(10) + shuffle at construction (env.py:693-694), then `_next_minibatch` with the epoch wrap's
    re-shuffle and the stable sort by instruction length (:723-735) -- one and a half epochs, sorted and unsorted."""
    items, gold, _, _ = world
    env, _, _ = VS.build_env(items, batch_size=gold['config']['batch'])     # (a fresh shuffle: construction seeds `random`)
    for k, want in enumerate(gold['minibatches']):
        env._next_minibatch(k % 2 == 0)
        assert [it['instr_id'] for it in env.batch] == want, k


def _same_adj(adj, rows, exact=True):
    assert [d['nextViewpointId'] for d in adj] == [r[0] for r in rows]
    assert [d['absViewIndex'] for d in adj] == [r[1] for r in rows]
    for d, r in zip(adj[1:], rows[1:]):
        if exact:
            assert d['rel_heading'] == r[2] and d['rel_elevation'] == r[3]
        else:
            assert abs(d['rel_heading'] - r[2]) <= 1e-12 and abs(d['rel_elevation'] - r[3]) <= 1e-12


def test_panorama_teacher_and_step_equal_the_reference_env(world):
    from speaker_follower_amd.env import WorldState, snapped_view
    items, gold, env, table = world
    n_adj = []
    for s in gold['states']:
        ws = WorldState(s['scan'], s['viewpoint'], s['heading'], s['elevation'])
        view, adj = env.panorama(ws)
        # the view the pose snaps to (MatterSim.cpp:339-367), also without a simulator call
        assert view == s['viewIndex'] == snapped_view(s['heading'], s['elevation'])
        _same_adj(adj, s['adj'])                                             # order, representative views, angles: bit for bit
        # the native batched sweep's tables (what the device holds), same state
        sid = table.row_of[(s['scan'], s['viewpoint'])] * 36 + view
        _same_adj(table.adj_loc_list(sid), s['adj'])
        # teacher (env.py:742-761 over networkx's all-pairs Dijkstra paths)
        assert env._teacher(ws, adj, s['goal']) == s['teacher']
        hop = table.hops(s['scan'], s['goal'])[table.row_of[(s['scan'], s['viewpoint'])] - table.base[s['scan']]]
        if s['teacher'] == 0:
            assert table.vp_of[hop][1] == s['viewpoint']
        else:
            assert table.vp_of[hop][1] == s['adj'][s['teacher']][0]
        # env.step (env.py:628-641 + 126-146): the pose every candidate leads to
        ob = dict(adj_loc_list=adj)
        for a, (vp2, h2, e2, view2) in enumerate(s['next']):
            w2 = env._step_one(ws, a, ob)
            assert w2.viewpointId == vp2
            if a == 0 or vp2 == s['viewpoint']:
                assert w2 is ws or w2 == ws                                  # stays: the pose object is returned as it came
                continue
            assert abs(w2.heading - h2) <= 1e-12 and abs(w2.elevation - e2) <= 1e-12
            assert snapped_view(w2.heading, w2.elevation) == view2 == s['adj'][a][1]
        n_adj.append(len(adj))
    assert len(n_adj) == gold['config']['n_states'] and max(n_adj) >= 9 and min(n_adj) >= 2


def test_gold_routes_equal_the_reference_env(world):
    """gold_obs_actions_and_instructions (env.py:823-854) over the whole split: visited viewpoints, view indices,
    headings, teacher actions, candidate counts -- from the dictionary env and from the index tables."""
    items, gold, _, table = world
    env, _, _ = VS.build_env(items, batch_size=gold['config']['batch'])
    S = gold['config']['max_steps']
    seen = {}
    for _ in range(9):
        path_obs, path_actions, enc = env.gold_obs_actions_and_instructions(S)
        n, rows = table.gold_routes(env.batch, S)
        first = np.concatenate(([0], np.cumsum(n)))
        for b, (obs, acts, e) in enumerate(zip(path_obs, path_actions, enc)):
            w = gold['routes'][obs[0]['instr_id']]
            assert [ob['viewpoint'] for ob in obs] == w['viewpoints']
            assert [ob['viewIndex'] for ob in obs] == w['views']
            assert [int(a) for a in acts] == w['actions'] and len(e) == w['n_tokens']
            assert [len(ob['adj_loc_list']) for ob in obs] == w['a_num']
            np.testing.assert_allclose([ob['heading'] for ob in obs], w['headings'], atol=1e-12)
            # index form (what the speaker's scoring consumes): one row per (state, action)
            r = rows[first[b]:first[b + 1]]
            assert int(n[b]) == len(acts)
            assert [int(v) for v in r[:, 1]] == w['views'][:len(acts)]
            assert [env.row_of[obs[0]['scan'] + '_' + v] for v in w['viewpoints'][:len(acts)]] == [int(x) for x in r[:, 0]]
            assert [bool(x) for x in r[:, 5]] == [a == 0 for a in acts]
            seen[obs[0]['instr_id']] = True
    assert len(seen) == 782
