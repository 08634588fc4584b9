"""CPU: the R2R index environment (N2) over the navigation-only MatterSim on committed
connectivity fixtures: panorama sweep, candidate ordering, teacher, transitions, index batches."""
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONN = os.path.join(ROOT, 'tests', 'golden', 'connectivity')
SCANS = ['YmJkqBEsHnH', 'gZ6f7yhEvPG', 'GdvgFV5R1Z5', '8194nk5LbLH']


@pytest.fixture(scope='module')
def envmod():
    from speaker_follower_amd.build import build_sim
    build_sim()
    from speaker_follower_amd import env
    return env


@pytest.fixture(scope='module')
def setup(envmod):
    graphs = {s: envmod.NavGraph(os.path.join(CONN, s + '_connectivity.json')) for s in SCANS}
    rng = np.random.default_rng(0)
    items = envmod.random_items(graphs, 24, rng)
    row_of, n = {}, 0
    for s, g in graphs.items():
        for v in g.ids:
            row_of[s + '_' + v] = n
            n += 1
    table = rng.random((n, 36, 16), dtype=np.float32)       # small feature dim: host logic only
    e = envmod.R2RIndexEnv(items, row_of, CONN, batch_size=8, host_table=table)
    return e, graphs, items


def test_panorama_sweep_finds_exactly_the_graph_neighbours(envmod, setup):
    e, graphs, _ = setup
    for scan, g in graphs.items():
        for vp in g.nodes()[:12]:
            for heading in (0.0, 1.3, 4.0):
                view, adj = e.panorama(envmod.WorldState(scan, vp, heading, 0))
                assert 12 <= view < 24                                         # horizon row after reset
                assert adj[0]['absViewIndex'] == -1 and adj[0]['nextViewpointId'] == vp
                assert {a['nextViewpointId'] for a in adj[1:]} == set(g.adj[vp])   # full 360 degree sweep
                hs = [abs(a['rel_heading']) for a in adj[1:]]
                assert hs == sorted(hs)                                        # env.py:221-222
                for a in adj[1:]:
                    assert 0 <= a['absViewIndex'] < 36
                    assert -math.pi <= a['rel_heading'] <= math.pi


def test_snapped_view_is_the_simulators_own_snapping(setup, envmod):
    """env.snapped_view (what the searches' start states use) against newEpisode + getState of the simulator: random poses,
    negative and multi-turn headings, and the exact half-way headings between two views (lround: away from zero)."""
    e = setup['env'] if isinstance(setup, dict) else setup[0]
    item = e.data[0]
    scan, vp = item['scan'], item['path'][0]
    rng = np.random.RandomState(3)
    headings = list(rng.uniform(-15.0, 15.0, 400)) + [k * math.pi / 12 for k in range(-30, 31)] + [0.0, 2 * math.pi, -2 * math.pi]
    elevations = [0.0, -0.6, 0.6, -math.pi / 12, math.pi / 12, 0.2617, -0.2619]
    n = 0
    for hd in headings:
        for el in (elevations if n % 7 == 0 else [0.0]):
            e.sim.newEpisode(scan, vp, hd, el)
            assert envmod.snapped_view(hd, el) == e.sim.getState().viewIndex, (hd, el)
        n += 1


def test_panorama_cache_hits(setup, envmod):
    e, graphs, _ = setup
    scan = SCANS[0]
    vp = graphs[scan].nodes()[0]
    n0 = len(e._pano)
    a = e.panorama(envmod.WorldState(scan, vp, 0.1, 0))
    n1 = len(e._pano)
    b = e.panorama(envmod.WorldState(scan, vp, 0.12, 0))               # same discretised heading
    assert len(e._pano) == n1 >= n0 and a is b


def test_teacher_walk_reaches_every_goal(setup):
    e, graphs, items = setup
    e.reset_epoch()
    path_obs, path_actions, enc = e.gold_obs_actions_and_instructions(10)
    assert len(path_obs) == 8
    for po, pa, item in zip(path_obs, path_actions, e.batch):
        assert len(po) == len(pa) + 1 and pa[-1] == 0                  # ends with stop
        assert po[0]['viewpoint'] == item['path'][0]
        assert po[-1]['viewpoint'] == item['path'][-1]
        g = graphs[item['scan']]
        walked = sum(g.adj[a['viewpoint']][b['viewpoint']] for a, b in zip(po[:-2], po[1:-1]))
        assert walked == pytest.approx(g.distance(item['path'][0], item['path'][-1]), rel=1e-6)
        for ob, a in zip(po, pa):
            assert ob['teacher'] == a and 0 <= a < len(ob['adj_loc_list'])


def test_step_transitions_match_simulator_navigation(setup, envmod):
    e, graphs, _ = setup
    ws = e.reset()
    obs = e.observe(ws)
    actions = [min(1, len(ob['adj_loc_list']) - 1) for ob in obs]
    nxt = e.step(ws, actions, obs)
    for w, a, ob, n in zip(ws, actions, obs, nxt):
        attr = ob['adj_loc_list'][a]
        if a == 0:
            assert n == w
            continue
        assert n.viewpointId == attr['nextViewpointId']
        # replay env.py:126-146 on the simulator: turn to the candidate's view, then move
        sim = e.sim
        sim.newEpisode(w.scanId, w.viewpointId, w.heading, w.elevation)
        st = sim.getState()
        dh = (attr['absViewIndex'] % 12 - st.viewIndex % 12 + 6) % 12 - 6
        for _ in range(abs(dh)):
            sim.makeAction(0, np.sign(dh), 0)
        de = attr['absViewIndex'] // 12 - st.viewIndex // 12
        for _ in range(abs(de)):
            sim.makeAction(0, 0, np.sign(de))
        st = sim.getState()
        assert st.viewIndex == attr['absViewIndex']
        idx = [l.viewpointId for l in st.navigableLocations].index(attr['nextViewpointId'])
        sim.makeAction(idx, 0, 0)
        st = sim.getState()
        assert st.location.viewpointId == n.viewpointId
        assert st.heading == pytest.approx(n.heading, abs=1e-9)
        assert st.elevation == pytest.approx(n.elevation, abs=1e-9)


def test_gold_index_batch_and_dense_observations_agree(setup):
    e, graphs, _ = setup
    e.reset_epoch()
    fb, path_obs, path_actions = e.gold_index_batch(10)
    S, B = fb.vp.shape
    assert B == 8 and S == max(len(a) for a in path_actions)
    lens = [len(i) for i in fb.instr]
    assert lens == sorted(lens, reverse=True)
    for b, (po, pa) in enumerate(zip(path_obs, path_actions)):
        for t in range(S):
            ob = po[t] if t < len(pa) else po[len(pa) - 1]
            assert fb.vp[t, b] == ob['vp_row'] and fb.view[t, b] == ob['viewIndex']
            assert fb.a_num[t, b] == len(ob['adj_loc_list'])
            assert fb.target[t, b] == (pa[t] if t < len(pa) else -1)
            emb = ob['action_embedding']
            for a in range(1, fb.a_num[t, b]):                         # dense row == indexed row
                np.testing.assert_array_equal(emb[a, :16], e.host_table[fb.vp[t, b], fb.cand_view[t, b, a]])
                np.testing.assert_allclose(emb[a, 16], math.sin(fb.cand_heading[t, b, a]), atol=1e-6)
            assert ob['feature'][0].shape == (36, 16 + 128)


def test_minibatching_cycles_through_the_data(setup):
    e, _, items = setup
    e.reset_epoch()
    seen = []
    for _ in range(3):
        e.reset()
        seen += [it['instr_id'] for it in e.batch]
    assert len(seen) == 24 and len(set(seen)) == 24                    # 24 items, batch 8: one epoch


def _fixture_env(scans):
    from speaker_follower_amd import env
    graphs = {s: env.NavGraph(os.path.join(CONN, s + '_connectivity.json')) for s in scans}
    items = env.random_items(graphs, 12, np.random.default_rng(3))
    row_of, n = {}, 0
    for s, g in graphs.items():
        for v in g.ids:
            row_of[s + '_' + v] = n
            n += 1
    return env.R2RIndexEnv(items, row_of, CONN, batch_size=12), n


def test_native_batched_sweep_is_bit_identical_to_the_per_state_sweep():
    """sim/sweep_py.cpp (the panorama sweep of env.py:149-224 for every state of a scan in one
    native call) against env.panorama_sweep driven from Python through the MatterSim binding: candidate
    order, next viewpoint, absViewIndex and the float64 angles, all 36 views of every viewpoint."""
    import math
    from speaker_follower_amd import env, sim
    from speaker_follower_amd.build import build_sim
    build_sim(verbose=False)
    sweep = sim.load_sweep().sweep_scan
    e, _ = _fixture_env(['17DRP5sb8fy', 'gZ6f7yhEvPG', 'YmJkqBEsHnH'])
    checked = 0
    for scan, g in e.graphs.items():
        nodes = [v for v, inc in zip(g.ids, g.included) if inc]
        a_num, nx, av, rh, re, ds = sweep(CONN, scan, nodes, env.IMAGE_W, env.IMAGE_H, math.radians(env.VFOV), 0)
        for r, v in enumerate(nodes):
            for view in range(36):
                got_view, adj = e.panorama(env.WorldState(scan, v, (view % 12) * env.ANGLE_INC,
                                                          (view // 12 - 1) * env.ANGLE_INC))
                h = view % 12
                assert got_view == view and a_num[r, h] == len(adj)
                for a, d in enumerate(adj[1:], 1):
                    assert nodes[nx[r, h, a]] == d['nextViewpointId'] and av[r, h, a] == d['absViewIndex']
                    assert rh[r, h, a] == d['rel_heading'] and re[r, h, a] == d['rel_elevation']     # bit-exact
                    assert ds[r, h, a] == d['distance']
                    checked += 1
    assert checked > 5000


def test_nav_table_rows_follow_the_sweep(tmp_path):
    """nav.NavTable (host side; the tables live on whatever device the store is on -- CPU here): per state
    a_num / next row / candidate view / sin-cos equal what env.observe would hand the agent."""
    import torch
    from speaker_follower_amd import env, nav, features
    from speaker_follower_amd.build import build_sim
    build_sim(verbose=False)
    e, n = _fixture_env(['gZ6f7yhEvPG', 'GdvgFV5R1Z5'])
    store = features.FeatureStore(np.zeros((n, 36, 8), np.float32), device='cpu')
    nt = nav.NavTable(e, store)
    assert nt.n_rows == sum(sum(g.included) for g in e.graphs.values())
    for r, (scan, v) in enumerate(nt.vp_of):
        for view in (0, 7, 13, 22, 35):
            _, adj = e.panorama(env.WorldState(scan, v, (view % 12) * env.ANGLE_INC, (view // 12 - 1) * env.ANGLE_INC))
            s = r * 36 + view
            assert int(nt.a_num[s]) == len(adj)
            assert int(nt.next_row[s, 0]) == r
            # the dictionary form rebuilt from the tables: the sweep's own list, key order and every float included
            rebuilt = nt.adj_loc_list(s)
            assert rebuilt == adj and [list(d) for d in rebuilt] == [list(d) for d in adj]
            for a, d in enumerate(adj[1:], 1):
                assert nt.vp_of[int(nt.next_row[s, a])] == (scan, d['nextViewpointId'])
                assert int(nt.cand_view[s, a]) == d['absViewIndex']
                want = features.cand_sincos(np.array([d['rel_heading']]), np.array([d['rel_elevation']]))[0]
                assert np.array_equal(nt.sincos[s, a].numpy(), want.astype(np.float32))
    # the teacher table: next hop toward a goal from every viewpoint of the scan
    scan = 'gZ6f7yhEvPG'
    goal = nt.vp_of[nt.base[scan]][1]
    hops = nt.hops(scan, goal)
    g = e.graphs[scan]
    for i, hrow in enumerate(hops):
        v = nt.vp_of[nt.base[scan] + i][1]
        p = g.path(v, goal)
        assert nt.vp_of[hrow][1] == (v if v == goal or p is None else p[1])


def test_full_r2r_geometry_table_regenerates_the_connectivity_directory(tmp_path):
    """data/r2r_connectivity.npz (tools/make_nav_geometry.py): all 90 scans / 10 567 included viewpoints; the
    regenerated files parse to graphs equal to the committed fixture files (same ids, flags, positions to the
    bit, edges and weights), and the native sweep runs over every scan."""
    import math
    from speaker_follower_amd import env, nav_data, sim
    from speaker_follower_amd.build import build_sim
    build_sim(verbose=False)
    geo = nav_data.load_geometry()
    assert len(geo) == 90 and sum(int(g['included'].sum()) for g in geo.values()) == 10567
    d = nav_data.connectivity_dir(out_dir=str(tmp_path / 'conn'))
    for f in sorted(os.listdir(CONN)):
        if not f.endswith('_connectivity.json'):
            continue
        a, b = env.NavGraph(os.path.join(CONN, f)), env.NavGraph(os.path.join(d, f))
        assert a.ids == b.ids and a.included == b.included and a.adj == b.adj
        assert all(np.array_equal(a.pos[k], b.pos[k]) for k in a.pos)
    sweep = sim.load_sweep().sweep_scan
    a_max, states = 0, 0
    for s, g in geo.items():
        nodes = [v for v, inc in zip(g['ids'], g['included']) if inc]
        a_num, nx, _, _, _, _ = sweep(d, s, nodes, env.IMAGE_W, env.IMAGE_H, math.radians(env.VFOV), 0)
        assert (nx >= 0).all() and (a_num >= 1).all()
        a_max = max(a_max, int(a_num.max()))
        states += a_num.size * 3
    assert states == 10567 * 36 and a_max == 14          # A max 14 over the 90 graphs


def test_gold_routes_from_the_tables_equal_the_environment_walk():
    """nav.NavTable.gold_routes (the speaker agent's minibatch in index form, table look-ups over the whole minibatch)
    against env.gold_obs_actions_and_instructions (env.py:823-848: the lock-step teacher walk over observation
    dictionaries): same actions, and per (state, action) the row the speaker's scoring forms from the dictionaries
    (speaker.py:68-121) -- feature row, view, the action's view, rel_heading / rel_elevation bit for bit, stop flag."""
    from speaker_follower_amd import nav, features
    from speaker_follower_amd.build import build_sim
    build_sim(verbose=False)
    e, n = _fixture_env(['gZ6f7yhEvPG', 'GdvgFV5R1Z5', 'YmJkqBEsHnH'])
    store = features.FeatureStore(np.zeros((n, 36, 8), np.float32), device='cpu')
    nt = nav.NavTable(e, store)
    for max_steps in (10, 3):                                  # (3: routes cut before their stop action)
        e.reset_epoch()
        path_obs, path_actions, _ = e.gold_obs_actions_and_instructions(max_steps)
        items = list(e.batch)
        counts, rows = nt.gold_routes(items, max_steps)
        assert counts.tolist() == [len(a) for a in path_actions]
        want = []
        for obs, actions in zip(path_obs, path_actions):
            for t, a in enumerate(actions):
                ob = obs[t]
                if a > 0:
                    d = ob['adj_loc_list'][a]
                    want.append((ob['vp_row'], ob['viewIndex'], d['absViewIndex'], d['rel_heading'], d['rel_elevation'], 0.0))
                else:
                    want.append((ob['vp_row'], ob['viewIndex'], 0, 0.0, 0.0, 1.0))
        assert np.array_equal(rows, np.array(want, np.float64))
        if max_steps == 10:
            assert all(a[-1] == 0 for a in path_actions) and len({len(a) for a in path_actions}) > 1
