"""Room-to-Room batch environment over the navigation-only MatterSim.

Own implementation of what tasks/R2R/env.py does around the simulator -- the 36-view panorama sweep
that enumerates navigable neighbours (env.py:149-224), navigation to a chosen candidate (:126-146),
the shortest-path teacher (:742-761), minibatching (:723-735) and observation assembly (:763-804)
-- with two differences that matter on MI355X:

  * the sweep result is a pure function of (scan, viewpoint, view index), so it is computed ONCE per
    state and cached (the reference re-walks 36 views with ~80 Python->C++ calls per sample per step);
  * observations are emitted in INDEX form (feature-table row, view index, per-candidate absolute
    view index + relative angles) which is what the HIP kernels gather from; the dense 2176-d numpy
    rows of the reference are built only on request (`dense=True`) for the dictionary-based agents.
"""
import heapq
import json
import math
import os
import random
from collections import namedtuple

import numpy as np

from . import sim as _sim
from .features import build_loc_table

WorldState = namedtuple('WorldState', ['scanId', 'viewpointId', 'heading', 'elevation'])   # env.py:227
ANGLE_INC = math.pi / 6.0
IMAGE_W, IMAGE_H, VFOV = 640, 480, 60          # env.py:289-291


class NavGraph:
    """One scan's connectivity graph: included viewpoints, unobstructed edges weighted by Euclidean
    distance (utils.py:26-51), shortest paths by Dijkstra."""

    def __init__(self, path):
        data = json.load(open(path))
        self.ids = [d['image_id'] for d in data]
        self.included = [d['included'] for d in data]
        self.pos = {d['image_id']: np.array([d['pose'][3], d['pose'][7], d['pose'][11]]) for d in data}
        self.adj = {}
        for i, a in enumerate(data):
            if not a['included']:
                continue
            for j, ok in enumerate(a['unobstructed']):
                if ok and data[j]['included']:
                    w = float(np.linalg.norm(self.pos[a['image_id']] - self.pos[data[j]['image_id']]))
                    self.adj.setdefault(a['image_id'], {})[data[j]['image_id']] = w
        self._sp = {}

    def nodes(self):
        return list(self.adj.keys())

    def _all_pairs(self):
        """Every source at once with scipy's Dijkstra (C): (node list, index, dist [n,n], pred [n,n]) -- or None when
        scipy is missing or a shortest path of this graph is not unique to the last bit (two predecessors of a node at
        exactly equal distance: then the per-source search below, whose relaxation order decides, stays the authority).
        The first pass over a new scan asked for ~150 Python searches (one per viewpoint, nav.NavTable.hops); this is one
        native call per scan.  Used by next_hops only."""
        if not hasattr(self, '_ap'):
            self._ap = None
            try:
                from scipy.sparse import csr_matrix
                from scipy.sparse.csgraph import dijkstra
                nodes = self.nodes()
                ix = {v: i for i, v in enumerate(nodes)}
                n = len(nodes)
                rows, cols, vals = [], [], []
                for u, nb in self.adj.items():
                    for v, w in nb.items():
                        rows.append(ix[u]); cols.append(ix[v]); vals.append(w)
                rows, cols, vals = np.array(rows), np.array(cols), np.array(vals, np.float64)
                if n and (vals > 0).all():
                    dist, pred = dijkstra(csr_matrix((vals, (rows, cols)), shape=(n, n)), directed=True,
                                          return_predecessors=True)
                    # unique to the last bit?  an edge (u, v) with dist[s, u] + w == dist[s, v] that is not v's predecessor
                    via = dist[:, rows] + vals[None, :]                      # [sources, edges]
                    tie = (via == dist[:, cols]) & (pred[:, cols] != rows[None, :]) & np.isfinite(via)
                    if not tie.any():
                        self._ap = (nodes, ix, dist, pred)
            except ImportError:
                pass
        return self._ap

    def shortest(self, src):
        """(dist, prev) from src to every reachable node."""
        if src not in self._sp:
            # (always the Python search: callers iterate these dictionaries, and their ORDER -- the order of discovery --
            # decides e.g. which items env.random_items draws; the all-pairs table serves next_hops only)
            dist, prev, heap = {src: 0.0}, {}, [(0.0, src)]
            while heap:
                d, u = heapq.heappop(heap)
                if d > dist.get(u, math.inf):
                    continue
                for v, w in self.adj.get(u, {}).items():
                    nd = d + w
                    if nd < dist.get(v, math.inf):
                        dist[v], prev[v] = nd, u
                        heapq.heappush(heap, (nd, v))
            self._sp[src] = (dist, prev)
        return self._sp[src]

    def next_hops(self, dst):
        """{node: the second node of path(node, dst)} for every node that has a path of at least one edge to dst -- the
        teacher's table (env.py:742-761) for a whole scan, straight from the all-pairs predecessor matrix when there is
        one (no per-source dictionaries), else from path()."""
        ap = self._all_pairs()
        if ap is None or dst not in ap[1]:
            out = {}
            for v in self.nodes():
                if v != dst:
                    p = self.path(v, dst)
                    if p is not None:
                        out[v] = p[1]
            return out
        nodes, ix, dist, pred = ap
        n, j = len(nodes), ix[dst]
        src = np.arange(n)
        ok = np.isfinite(dist[:, j]) & (src != j)
        x = np.full(n, j)
        for _ in range(n):                                   # walk back from dst until the predecessor is the source
            p = pred[src, x]
            more = ok & (p != src) & (p >= 0)
            if not more.any():
                break
            x = np.where(more, p, x)
        return {nodes[i]: nodes[x[i]] for i in np.flatnonzero(ok)}

    def path(self, src, dst):
        dist, prev = self.shortest(src)
        if dst not in dist:
            return None
        out = [dst]
        while out[-1] != src:
            out.append(prev[out[-1]])
        return out[::-1]

    def distance(self, src, dst):
        return self.shortest(src)[0].get(dst, math.inf)


def make_sim(nav_graph_path):
    """env.py:239-246: rendering off, discretized 12x3 views, 640x480, 60 degree vfov."""
    ms = _sim.load()
    s = ms.Simulator()
    s.setRenderingEnabled(False)
    s.setDiscretizedViewingAngles(True)
    s.setCameraResolution(IMAGE_W, IMAGE_H)
    s.setCameraVFOV(math.radians(VFOV))
    s.setNavGraphPath(nav_graph_path)
    s.init()
    return s


def _canonical_angle(x):
    return x - 2 * math.pi * round(x / (2 * math.pi))                    # env.py:108-110


def panorama_sweep(sim):
    """env.py:149-224: look down, walk the 36 discrete views, keep for every neighbour the view in
    which it is closest to the image centre; candidate 0 = stop, the rest sorted by |rel_heading|.
    Returns (viewIndex, adj_loc_list); the simulator is left in its initial view."""
    st = sim.getState()
    init_view = st.viewIndex
    delta = -(st.viewIndex // 12)
    for _ in range(abs(delta)):
        sim.makeAction(0, 0, -1)
    adj = {}
    for rel in range(36):
        base_h = (rel % 12) * ANGLE_INC
        base_e = (rel // 12 - 1) * ANGLE_INC
        st = sim.getState()
        for loc in st.navigableLocations[1:]:
            dist = math.sqrt(loc.rel_heading ** 2 + loc.rel_elevation ** 2)
            if loc.viewpointId not in adj or dist < adj[loc.viewpointId]['distance']:
                adj[loc.viewpointId] = dict(absViewIndex=st.viewIndex, nextViewpointId=loc.viewpointId,
                                            rel_heading=_canonical_angle(base_h + loc.rel_heading),
                                            rel_elevation=base_e + loc.rel_elevation, distance=dist)
        if (rel + 1) % 12 == 0:
            sim.makeAction(0, 1, 1)
        else:
            sim.makeAction(0, 1, 0)
    for _ in range(abs(-2 - delta)):
        sim.makeAction(0, 0, 1 if (-2 - delta) > 0 else -1)
    st = sim.getState()
    assert st.viewIndex == init_view
    stop = dict(absViewIndex=-1, nextViewpointId=st.location.viewpointId, rel_heading=0.0,
                rel_elevation=0.0, distance=0.0)
    return st.viewIndex, [stop] + sorted(adj.values(), key=lambda x: abs(x['rel_heading']))


def snapped_view(heading, elevation=0.0):
    """The view index the simulator snaps a continuous pose to (MatterSim.cpp:339-367 with discretized viewing angles:
    heading to the nearest of 12 steps of 30 degrees, halves away from zero; elevation to -30 / 0 / +30 degrees), as
    sim/mattersim_nav.cpp's setHeadingElevation does -- without a simulator call (tests/test_env.py checks the two)."""
    h = math.fmod(heading, 2.0 * math.pi)
    while h < 0.0:
        h += 2.0 * math.pi
    x = h / (2.0 * math.pi / 12)
    step = int(x)
    if x - step >= 0.5:                                              # lround
        step += 1
    if step == 12:
        step = 0
    if elevation < -ANGLE_INC / 2.0:
        return step
    if elevation > ANGLE_INC / 2.0:
        return step + 24
    return step + 12


class R2RIndexEnv:
    """R2RBatch (env.py:664-854) over index-form observations.

    items: dicts with 'scan', 'path' (viewpoint ids, path[0] = start, path[-1] = goal), 'heading',
    'instr_id', 'instr_encoding'.  row_of: 'scan_viewpoint' -> feature-table row.  host_table
    ([n,36,img] numpy, optional) enables dense observations for the dictionary-based agents."""

    def __init__(self, items, row_of, nav_graph_path, batch_size=100, seed=10, host_table=None,
                 loc=128, scans=None):
        self.data = list(items)
        self.row_of = row_of
        self.nav_graph_path = nav_graph_path
        self.batch_size = batch_size
        self.beam_size = 1
        self.host_table = host_table
        self.loc_table = build_loc_table(36, loc)
        self.loc = loc
        self.tokenizer = None
        self.image_features_list = [None]
        random.seed(seed)                                                     # env.py:693
        random.shuffle(self.data)
        self.ix = 0
        self.sim = make_sim(nav_graph_path)
        # scans: graphs to hold beyond those of `items` (a nav.NavTable covers exactly self.graphs)
        self.graphs = {s: NavGraph(os.path.join(nav_graph_path, s + '_connectivity.json'))
                       for s in sorted({it['scan'] for it in self.data} | set(scans or ()))}
        self._pano = {}

    # ---- minibatching (env.py:723-740)
    def _next_minibatch(self, sort):
        batch = self.data[self.ix:self.ix + self.batch_size]
        if len(batch) < self.batch_size:
            random.shuffle(self.data)
            self.ix = self.batch_size - len(batch)
            batch += self.data[:self.ix]
        else:
            self.ix += self.batch_size
        if sort:
            batch = sorted(batch, key=lambda it: len(it['instr_encoding']), reverse=True)
        self.batch = batch

    def peek_next_minibatch(self, sort):
        """The minibatch the next `reset` will draw, WITHOUT drawing it -- or None where that draw wraps the epoch (it
        reshuffles the data with `random`, env.py:601-614: not something to do ahead of time)."""
        batch = self.data[self.ix:self.ix + self.batch_size]
        if len(batch) < self.batch_size:
            return None
        return sorted(batch, key=lambda it: len(it['instr_encoding']), reverse=True) if sort else batch

    def reset_epoch(self):
        self.ix = 0

    def reset(self, sort=False, beamed=False, load_next_minibatch=True):
        """env.py:814-819 / :601-614.  beamed: one list of world states per instance (the search
        procedures' interface)."""
        if load_next_minibatch:
            self._next_minibatch(sort)
        ws = [WorldState(it['scan'], it['path'][0], it['heading'], 0) for it in self.batch]
        return [[w] for w in ws] if beamed else ws

    def set_beam_size(self, beam_size, force_reload=False):
        """env.py:700-709.  The reference keeps batch x beam simulators; here ONE simulator and
        the sweep cache serve any number of states, so this only records the size."""
        self.beam_size = beam_size

    def start_view(self, ws):
        """The discrete view index newEpisode snaps a pose to (env.py:814-819), without the panorama sweep."""
        return snapped_view(ws.heading, ws.elevation)

    # ---- cached panorama sweep
    def panorama(self, ws):
        self.sim.newEpisode(ws.scanId, ws.viewpointId, ws.heading, ws.elevation)
        view = self.sim.getState().viewIndex
        key = (ws.scanId, ws.viewpointId, view)
        if key not in self._pano:
            self._pano[key] = panorama_sweep(self.sim)
        return self._pano[key]

    def _teacher(self, ws, adj, goal):
        """env.py:742-761."""
        if ws.viewpointId == goal:
            return 0
        nxt = self.graphs[ws.scanId].path(ws.viewpointId, goal)[1]
        for n, a in enumerate(adj):
            if a['nextViewpointId'] == nxt:
                return n
        raise RuntimeError('next viewpoint %s not among the candidates of %s' % (nxt, ws.viewpointId))

    def observe(self, world_states, beamed=False, include_teacher=True, dense=None):
        dense = (self.host_table is not None) if dense is None else dense
        if beamed:                                                            # env.py:766-799, nested
            return [[self._observe_one(ws, item, include_teacher, dense) for ws in beam]
                    for beam, item in zip(world_states, self.batch)]
        return [self._observe_one(ws, item, include_teacher, dense)
                for ws, item in zip(world_states, self.batch)]

    def _observe_one(self, ws, item, include_teacher, dense, view=None):
        """`view`: the state's view index when the caller knows it (a discretised pose): the sweep cache is then
        consulted without waking the simulator."""
        hit = self._pano.get((ws.scanId, ws.viewpointId, view)) if view is not None else None
        view, adj = hit if hit is not None else self.panorama(ws)
        # heading / elevation are the SIMULATOR's (env.py:783-784: state.heading after newEpisode snapped the pose to the
        # discrete view, MatterSim.cpp:339-367), not the world state's: an item's start heading is continuous
        ob = dict(instr_id=item['instr_id'], scan=ws.scanId, viewpoint=ws.viewpointId,
                  viewIndex=view, heading=(view % 12) * ANGLE_INC, elevation=(view // 12 - 1) * ANGLE_INC,
                  adj_loc_list=adj, vp_row=self.row_of[ws.scanId + '_' + ws.viewpointId],
                  instr_encoding=item['instr_encoding'], instructions=item.get('instructions', ''))
        if include_teacher:
            ob['teacher'] = self._teacher(ws, adj, item['path'][-1])
        if dense:
            feats = self.host_table[ob['vp_row']]
            ob['feature'] = [np.concatenate((feats, self.loc_table[view]), axis=-1)]      # env.py:773
            ob['action_embedding'] = self._action_embedding(adj, feats)                   # env.py:774
        return ob

    def _action_embedding(self, adj, feats):
        g = self.loc // 4
        emb = np.zeros((len(adj), feats.shape[-1] + self.loc), np.float32)
        for a, d in enumerate(adj):
            if a == 0:
                continue
            emb[a, :feats.shape[-1]] = feats[d['absViewIndex']]
            le = emb[a, feats.shape[-1]:]
            le[0:g], le[g:2 * g] = math.sin(d['rel_heading']), math.cos(d['rel_heading'])
            le[2 * g:3 * g], le[3 * g:] = math.sin(d['rel_elevation']), math.cos(d['rel_elevation'])
        return emb

    def step(self, world_states, actions, last_obs, beamed=False):
        """env.py:628-641 + :126-146: turning to the candidate's view and stepping leaves the agent at
        the neighbour, facing that view's heading / elevation; action 0 stays."""
        if beamed:
            return [[self._step_one(ws, a, ob) for ws, a, ob in zip(wl, al, ol)]
                    for wl, al, ol in zip(world_states, actions, last_obs)]
        return [self._step_one(ws, a, ob) for ws, a, ob in zip(world_states, actions, last_obs)]

    @staticmethod
    def _step_one(ws, a, ob):
        attr = ob['adj_loc_list'][int(a)]
        if int(a) == 0 or attr['nextViewpointId'] == ws.viewpointId:
            return ws
        v = attr['absViewIndex']
        return WorldState(ws.scanId, attr['nextViewpointId'], (v % 12) * ANGLE_INC,
                          (v // 12 - 1) * ANGLE_INC)

    def shortest_paths_to_goals(self, starting_world_states, max_steps):
        """env.py:823-848."""
        ws = starting_world_states
        obs = self.observe(ws)
        all_obs, all_actions = [[ob] for ob in obs], [[] for _ in obs]
        ended = np.array([False] * len(obs))
        for _ in range(max_steps):
            actions = [ob['teacher'] for ob in obs]
            ws = self.step(ws, actions, obs)
            obs = self.observe(ws)
            for i, ob in enumerate(obs):
                if not ended[i]:
                    all_obs[i].append(ob)
            for i, a in enumerate(actions):
                if not ended[i]:
                    all_actions[i].append(a)
                    if a == 0:
                        ended[i] = True
            if ended.all():
                break
        return all_obs, all_actions

    def gold_obs_actions_and_instructions(self, max_steps, load_next_minibatch=True):
        ws = self.reset(load_next_minibatch=load_next_minibatch)
        path_obs, path_actions = self.shortest_paths_to_goals(ws, max_steps)
        return path_obs, path_actions, [obs[0]['instr_encoding'] for obs in path_obs]

    def gold_index_batch(self, max_steps, a_max=None, sort=True):
        """The current minibatch's teacher paths as a synth.FollowerBatch (index form) for
        FollowerEngine: step t of sample b = observation t along its shortest path; after the stop
        action the last observation repeats with target -1 (follower.py:376-381)."""
        from .synth import FollowerBatch
        ws = self.reset(sort=sort)
        path_obs, path_actions = self.shortest_paths_to_goals(ws, max_steps)
        B = len(path_obs)
        S = max(len(a) for a in path_actions)
        A = a_max or max(len(ob['adj_loc_list']) for po in path_obs for ob in po)
        fb = FollowerBatch(instr=[po[0]['instr_encoding'] for po in path_obs],
                           vp=np.zeros((S, B), np.int32), view=np.zeros((S, B), np.int32),
                           a_num=np.full((S, B), 1, np.int32), cand_view=np.zeros((S, B, A), np.int32),
                           cand_heading=np.zeros((S, B, A), np.float32),
                           cand_elevation=np.zeros((S, B, A), np.float32),
                           target=np.full((S, B), -1, np.int64), a_max=A)
        for b, (po, pa) in enumerate(zip(path_obs, path_actions)):
            for t in range(S):
                ob = po[min(t, len(pa) - 1)] if t >= len(pa) else po[t]
                adj = ob['adj_loc_list']
                fb.vp[t, b], fb.view[t, b], fb.a_num[t, b] = ob['vp_row'], ob['viewIndex'], len(adj)
                for a, d in enumerate(adj[1:], 1):
                    fb.cand_view[t, b, a] = d['absViewIndex']
                    fb.cand_heading[t, b, a] = d['rel_heading']
                    fb.cand_elevation[t, b, a] = d['rel_elevation']
                if t < len(pa):
                    fb.target[t, b] = pa[t]
        return fb, path_obs, path_actions


def random_items(graphs, n, rng, min_hops=2, max_hops=5, vocab=991, min_len=5, max_len=30):
    """Synthetic R2R items on real connectivity graphs: start viewpoint, goal 2..5 hops away along a
    shortest path, random heading, random instruction tokens."""
    items = []
    scans = sorted(graphs)
    while len(items) < n:
        scan = scans[int(rng.integers(len(scans)))]
        g = graphs[scan]
        nodes = g.nodes()
        if len(nodes) < 3:
            continue
        src = nodes[int(rng.integers(len(nodes)))]
        dist, prev = g.shortest(src)
        cands = [v for v in dist if v != src]
        if not cands:
            continue
        dst = cands[int(rng.integers(len(cands)))]
        path = g.path(src, dst)
        if not (min_hops <= len(path) - 1 <= max_hops):
            continue
        items.append(dict(scan=scan, path=path, heading=float(rng.uniform(0, 2 * math.pi)),
                          instr_id='%d_0' % len(items), path_id=len(items),
                          instr_encoding=rng.integers(4, vocab, size=int(rng.integers(min_len, max_len + 1))
                                                      ).astype(np.int64)))
    return items
