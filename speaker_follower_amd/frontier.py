"""Search over a struct-of-arrays frontier.

The reference's search procedures (follower.py:541-980, speaker.py:211-318) keep every hypothesis as a
namedtuple that points at its parent, world states as tuples, per-instance dictionaries keyed by world state,
and walk the simulator from Python for every successor.  Their RESULTS are pinned by the golden files
(tests/golden/g7_search*.json, produced by the reference itself); this module reaches the same results in this
repository's own form:

  * a hypothesis is a ROW of `Hypotheses` (parent pointer, instance, state id, last action, depth, float32
    score, row of its h / c / attention in the device pool): nothing is copied when a hypothesis is extended,
    lineages are parent-pointer walks over integer arrays;
  * a world state is an INTEGER: state id = nav row * 36 + view index (nav.NavTable: candidates, next states,
    view indices and sin/cos of every (viewpoint, view) tabulated once); only an episode's start pose, whose
    heading is continuous, gets an id of its own.  env.step / env.observe are table look-ups
    (`next_row[s, a]`, `cand_view[s, a]`), done for ALL successors of ALL instances at once;
  * the per-instance dictionaries of the state-factored search are dense [instance, key] tables (best
    hypothesis per state, expanded flag) plus an insertion-ordered entry list, updated and queried with
    numpy over the whole minibatch: one lexsort orders the successors the way the reference's per-instance
    sorted() does, one more picks each instance's next states to expand;
  * one flat decoder step per iteration on the device (search.FlatDecoder), its inputs gathered by fancy
    indexing from the host copies of the same tables the device holds.

Observation dictionaries (the reference's result format carries them: rational_follower.py:79-82 feeds them to
the speaker) are materialised once per distinct state, and only when a caller READS them: a result dictionary
is a `Candidate` whose 'observations' / 'trajectory' are built on first access (the speaker scores the routes in
index form: `_trajectories`' hook).
"""
from collections.abc import Sequence

import numpy as np
import torch

from .env import ANGLE_INC, WorldState
from .lazydict import LazyDict
from .runtime import gc_paused as _gc_paused

V = 36
F32 = np.float32


class Hypotheses:
    """Append-only struct-of-arrays of search hypotheses."""
    INT_FIELDS = ('parent', 'inst', 'sid', 'key', 'action', 'count', 'pool')

    @classmethod
    def from_arrays(cls, parent, inst, sid, key, action, count, pool, score, start_pose):
        """The finished table of the native bookkeeping (sim/frontier_core.cpp: StateFactored.hypotheses)."""
        t = cls.__new__(cls)
        t.n = t.cap = len(parent)
        t.parent, t.inst, t.sid, t.key, t.action, t.count, t.pool = parent, inst, sid, key, action, count, pool
        t.score, t.start_pose = score, start_pose.astype(bool)
        return t

    def __init__(self, cap=8192):
        self.n = 0
        self.cap = cap
        for f in self.INT_FIELDS:
            setattr(self, f, np.zeros(cap, np.int64))
        self.score = np.zeros(cap, F32)          # cumulative log-probability, every partial sum rounded to fp32
        self.start_pose = np.zeros(cap, bool)    # still in the episode's start pose (continuous heading)

    def append(self, score, start_pose, **ints):
        m = len(score)
        if self.n + m > self.cap:
            self.cap = max(2 * self.cap, self.n + m)
            for f in self.INT_FIELDS + ('score', 'start_pose'):
                old = getattr(self, f)
                new = np.zeros(self.cap, old.dtype)
                new[:self.n] = old[:self.n]
                setattr(self, f, new)
        ids = np.arange(self.n, self.n + m)
        for f in self.INT_FIELDS:
            getattr(self, f)[ids] = ints[f]
        self.score[ids] = score
        self.start_pose[ids] = start_pose
        self.n += m
        return ids

    def lineage(self, node):
        """node, parent(node), ..., root."""
        out = []
        while node >= 0:
            out.append(int(node))
            node = self.parent[node]
        return out


class StateSpace:
    """The minibatch's view of the navigation tables: start poses, state keys, observations on demand."""

    def __init__(self, env, nav, items, key_fields=4):
        self.env, self.nav, self.items = env, nav, items
        self.h = nav.host
        B = len(items)
        self.base_row = np.array([nav.base[it['scan']] for it in items], np.int64)
        self.rows = np.array([nav.scan_rows[it['scan']] for it in items], np.int64)
        root_row = np.array([nav.row_of[(it['scan'], it['path'][0])] for it in items], np.int64)
        root_view = np.array([env.start_view(WorldState(it['scan'], it['path'][0], it['heading'], 0))
                              for it in items], np.int64)         # newEpisode snaps the heading (env.py:814-819)
        self.root_sid = root_row * V + root_view
        # keys: what the reference's `world_state[0:first_n_ws_key]` distinguishes (follower.py:722, 843)
        self.key_fields = key_fields
        per = {4: V, 3: 12, 2: 1, 1: 0}[key_fields]
        self.n_keys = int((self.rows * per).max()) + 2
        self.root_key = np.full(B, self.n_keys - 1, np.int64) if key_fields >= 3 else self.key_of(self.root_sid, np.arange(B))
        self._obs = {}

    def key_of(self, sid, inst):
        local_row = sid // V - self.base_row[inst]
        if self.key_fields == 4:
            return local_row * V + sid % V
        if self.key_fields == 3:
            return local_row * 12 + sid % 12
        if self.key_fields == 2:
            return local_row
        return np.zeros_like(sid)

    def successors(self, sid, action):
        """env.step as a table look-up (env.py:126-146, 628-641): (next state id, stays in place)."""
        nxt = self.h['next_row'][sid, action]
        stay = (action == 0) | (nxt == sid // V)
        return np.where(stay, sid, nxt * V + self.h['cand_view'][sid, action]), stay

    def world_state(self, inst, sid, start_pose):
        it = self.items[inst]
        if start_pose:
            return WorldState(it['scan'], it['path'][0], it['heading'], 0)
        view = int(sid) % V
        return WorldState(it['scan'], self.nav.vp_of[int(sid) // V][1], (view % 12) * ANGLE_INC,
                          (view // 12 - 1) * ANGLE_INC)

    def observation(self, inst, sid, start_pose):
        """The env's observation dictionary of a state (env.py:763-804), built once per distinct state."""
        k = (int(inst), int(sid), bool(start_pose))
        ob = self._obs.get(k)
        if ob is None:
            ob = self._obs[k] = self._build(k)
        return ob

    def _build(self, k):
        env, nav = self.env, self.nav
        b, s, sp = k
        it = self.items[b]
        view = s % V
        scan = it['scan']
        # (the observation carries the simulator's snapped pose, env.py:783-784 -- for the start pose too; only the WORLD
        # STATE of the start keeps the item's continuous heading, world_state())
        vp, heading, elevation = nav.vp_of[s // V][1], (view % 12) * ANGLE_INC, (view // 12 - 1) * ANGLE_INC
        hit = env._pano.get((scan, vp, view))
        if hit is None:                                             # the candidate list from the tables, not a sweep
            hit = env._pano[(scan, vp, view)] = (view, nav.adj_loc_list(s))
        if env.host_table is None:                                  # index-form observations: the dictionary is built inline
            return dict(instr_id=it['instr_id'], scan=scan, viewpoint=vp, viewIndex=view, heading=heading,
                        elevation=elevation, adj_loc_list=hit[1], vp_row=env.row_of[scan + '_' + vp],
                        instr_encoding=it['instr_encoding'], instructions=it.get('instructions', ''))
        return env._observe_one(WorldState(scan, vp, heading, elevation), it, False, True, view=view)


class Hyp:
    """Read-only view of one row of `Hypotheses` with the attribute names the callers of the reference's
    InferenceState use (world_state, observation, score, last_action, prev_inference_state, ...)."""
    __slots__ = ('t', 'space', 'node')

    def __init__(self, t, space, node):
        self.t, self.space, self.node = t, space, int(node)

    @property
    def prev_inference_state(self):
        p = self.t.parent[self.node]
        return None if p < 0 else Hyp(self.t, self.space, p)

    @property
    def world_state(self):
        n = self.node
        return self.space.world_state(self.t.inst[n], self.t.sid[n], self.t.start_pose[n])

    @property
    def observation(self):
        n = self.node
        return self.space.observation(self.t.inst[n], self.t.sid[n], self.t.start_pose[n])

    score = property(lambda self: float(self.t.score[self.node]))
    last_action = property(lambda self: int(self.t.action[self.node]))
    action_count = property(lambda self: int(self.t.count[self.node]))
    last_alpha = property(lambda self: None if self.t.parent[self.node] < 0 else int(self.t.pool[self.node]))
    h_t = c_t = property(lambda self: int(self.t.pool[self.node]))

    def __eq__(self, other):
        return isinstance(other, Hyp) and other.node == self.node and other.t is self.t

    def __hash__(self):
        return hash(self.node)


def _ragged_arange(counts):
    """[0..c0-1, 0..c1-1, ...] and the index of the owner of each element."""
    owner = np.repeat(np.arange(len(counts)), counts)
    starts = np.cumsum(counts) - counts
    return np.arange(len(owner)) - starts[owner], owner


def _first_k_per_group(group_sorted, k):
    """Mask of the first k elements of every run of equal values in an already grouped array."""
    if len(group_sorted) == 0:
        return np.zeros(0, bool)
    new = np.r_[True, group_sorted[1:] != group_sorted[:-1]]
    start = np.flatnonzero(new)
    rank = np.arange(len(group_sorted)) - start[np.cumsum(new) - 1]
    return rank < k


def _step_inputs(space, t, frontier):
    """Index-form inputs of one flat decoder step for the hypotheses `frontier` (everything a fancy index)."""
    h = space.h
    sid = t.sid[frontier]
    par = t.parent[frontier]
    has_u = par >= 0                                   # roots start from u_begin = zeros (model.py:368)
    psid = t.sid[np.maximum(par, 0)]
    act = np.where(has_u, t.action[frontier], 0)
    return dict(vp=h['feat_row'][sid // V], view=sid % V, a_num=h['a_num'][sid], cand_view=h['cand_view'][sid],
                sincos=h['sincos'][sid], hrow=t.pool[frontier], crow=t.inst[frontier], has_u=has_u,
                u_vp=h['feat_row'][psid // V], u_view=h['cand_view'][psid, act], u_sincos=h['sincos'][psid, act]), sid


def _lineage_matrix(t, nodes, depth):
    """[len(nodes), depth + 1] parent-pointer chains (column 0 = the node, -1 beyond the root) and their lengths."""
    L = np.full((len(nodes), depth + 1), -1, np.int64)
    cur = np.asarray(nodes, np.int64)
    for d in range(depth + 1):
        L[:, d] = cur
        cur = np.where(cur >= 0, t.parent[np.maximum(cur, 0)], -1)
    return L, (L >= 0).sum(1)


class _Routes:
    """What the result dictionaries of one search are made from, kept so that their expensive fields -- the observation
    dictionaries and the pose tuples of every route -- are built when somebody reads them (rational_follower.py:67-69
    hands the observation lists to the speaker, which scores them in index form, and then deletes them; only the 64
    chosen routes of ~2 500 ever show their trajectory)."""

    def __init__(self, t, space, L, ln):
        self.space = space
        self.inst, self.sid, self.start = t.inst[L], t.sid[L], t.start_pose[L]      # (copies: [routes, depth + 1])
        self.lens = ln.tolist()
        self._obs = [None] * len(self.lens)

    def states(self, i):
        m = self.lens[i]
        return zip(self.inst[i, :m].tolist(), self.sid[i, :m].tolist(), self.start[i, :m].tolist())

    def observations(self, i):
        obs = self._obs[i]
        if obs is None:
            one = self.space.observation
            obs = self._obs[i] = [one(b, s, p) for b, s, p in self.states(i)]
        return obs

    def trajectory(self, i):
        """(viewpoint, heading, elevation) per state (follower.py:700): the pose fields of `observations`, without them."""
        vp_of, out = self.space.nav.vp_of, []
        for b, s, p in self.states(i):
            view = s % V
            out.append((vp_of[s // V][1], (view % 12) * ANGLE_INC, (view // 12 - 1) * ANGLE_INC))
        return out


class RouteObservations(Sequence):
    """The observation list of one candidate route: a read-only sequence whose dictionaries are built on first access."""
    __slots__ = ('_routes', '_i')

    def __init__(self, routes, i):
        self._routes, self._i = routes, i

    def __len__(self):
        return self._routes.lens[self._i]

    def __getitem__(self, k):
        return self._routes.observations(self._i)[k]

    def __iter__(self):
        return iter(self._routes.observations(self._i))

    def __eq__(self, other):
        return list(self) == list(other) if isinstance(other, (list, tuple, RouteObservations)) else NotImplemented

    def __repr__(self):
        return 'RouteObservations(%d states)' % len(self)

    def __reduce__(self):                      # (pickled / deep-copied as the plain list it stands for)
        return list, (list(self),)

    @property
    def instr_id(self):
        return self._routes.space.items[int(self._routes.inst[self._i, 0])]['instr_id']


class Candidate(LazyDict):
    """One result dictionary (follower.py:694-716 / 953-975).  'trajectory' and 'observations' are made when first read
    ('observations' as a RouteObservations); every way of reading that a plain dict offers sees them."""
    __slots__ = ('_routes', '_i')

    def __init__(self, routes, i, fields):
        LazyDict.__init__(self, fields, ('trajectory', 'observations'))
        self._routes, self._i = routes, i

    def _make(self, key):
        return self._routes.trajectory(self._i) if key == 'trajectory' else RouteObservations(self._routes, self._i)


def _trajectories(fd, t, space, completed_lists, depth, hook=None, fp32_steps=False):
    """Result dictionaries (follower.py:694-716 / 953-975) of the final hypotheses of every instance.
    `hook(n, rows, instructions)` (Seq2SeqSpeaker.route_scores_hook): called with the routes in index form -- per
    route its number of steps, per step (feature row, view index, the action's view, rel_heading, rel_elevation,
    is_stop), per route its instruction -- once the device is idle and BEFORE the dictionaries are built; what it
    returns is called with the routes' observation lists (RouteObservations: nothing is built for that)."""
    flat = np.array([n for lst in completed_lists for n in lst], np.int64)
    L, ln = _lineage_matrix(t, flat, depth)
    # root first: column j of every row = the j-th hypothesis of the path (columns >= its length: junk, sliced off)
    L = np.take_along_axis(L, np.maximum(ln[:, None] - 1 - np.arange(depth + 1)[None, :], 0), axis=1)
    act, sc, pool = t.action[L], t.score[L].astype(np.float64), t.pool[L]
    live = np.arange(depth + 1)[None, :] < ln[:, None]
    rows, inv = np.unique(pool[:, 1:][live[:, 1:]], return_inverse=True)      # every non-root hypothesis on a path
    att_rows = np.empty(len(rows), object)
    att_rows[:] = fd.attention_rows(rows.tolist())
    att = np.empty(pool.shape, object)
    att[:, 1:][live[:, 1:]] = att_rows[inv]
    bind = None
    if hook is not None:
        # step k of a route: the state of its k-th hypothesis, the action that led to the (k+1)-th
        h = space.h
        step = live[:, 1:]
        sid_k, a_k = t.sid[L[:, :-1]][step], act[:, 1:][step]              # route-major
        move = a_k > 0
        a_c = np.where(move, a_k, 0)
        rows = np.stack((h['feat_row'][sid_k // V], sid_k % V, np.where(move, h['cand_view'][sid_k, a_c], 0),
                         np.where(move, h['heading'][sid_k, a_c], 0.0), np.where(move, h['elevation'][sid_k, a_c], 0.0),
                         (~move).astype(np.float64)), axis=1).astype(np.float64)
        items = space.items
        bind = hook(ln - 1, rows, [items[b]['instr_encoding'] for b in t.inst[L[:, 0]].tolist()])
    routes = _Routes(t, space, L, ln)
    # per-action scores = differences of the cumulative ones: the reference's beam search holds them as Python floats
    # (follower.py:636: float64 differences of fp32-rounded sums), its state-factored search as fp32 tensors
    # (follower.py:851, :46: the difference is rounded to fp32)
    step_sc = (t.score[L][:, 1:] - t.score[L][:, :-1]).astype(np.float64) if fp32_steps else sc[:, 1:] - sc[:, :-1]
    # (whole matrices to nested Python lists ONCE; a candidate's fields are then plain list slices)
    act_l, step_l, att_l = act.tolist(), step_sc.tolist(), att.tolist()
    last_sc = sc[np.arange(len(flat)), ln - 1].tolist()
    lens = routes.lens
    items = space.items
    inst0 = t.inst[L[:, 0]].tolist()
    out, i = [], 0
    for lst in completed_lists:
        assert lst
        cands = []
        for _ in lst:
            m = lens[i]
            it = items[inst0[i]]
            cands.append(Candidate(routes, i, {
                'instr_id': it['instr_id'], 'instr_encoding': it['instr_encoding'], 'actions': act_l[i][1:m],
                'score': last_sc[i], 'scores': step_l[i][:m - 1], 'attentions': att_l[i][1:m]}))
            i += 1
        out.append(cands)
    if bind is not None:
        bind([c['observations'] for cands in out for c in cands])
    return out


class HypList:
    """A list of hypotheses held as node ids; `Hyp` views are made on access."""

    def __init__(self, t, space, nodes):
        self.t, self.space, self.nodes = t, space, list(nodes)

    def __len__(self):
        return len(self.nodes)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [Hyp(self.t, self.space, n) for n in self.nodes[i]]
        return Hyp(self.t, self.space, self.nodes[i])

    def __iter__(self):
        return (Hyp(self.t, self.space, n) for n in self.nodes)

    def __add__(self, other):
        return list(self) + list(other)


def _setup_space(agent, load_next_minibatch, key_fields=4, graph_cap=0):
    """env.reset + the minibatch's state space + the encoder pass + the decoder-step object: search.FlatDecoder
    (host-issued launches over any number of states), or -- graph_cap > 0 -- the agent's search.GraphStep of that
    capacity (one hipGraph replay per iteration)."""
    from . import nav, search
    env = agent.env
    search._require_store(agent)
    env.reset(sort=True, beamed=True, load_next_minibatch=load_next_minibatch)
    items = list(env.batch)
    table = nav.table_for(env, agent.store)
    space = StateSpace(env, table, items, key_fields)
    ctx, seq_mask, h_t, c_t = search._encode_items(agent, items)
    if graph_cap:
        fd = search.graph_step_for(agent, table, len(items), graph_cap)
        fd.load(ctx, seq_mask, h_t, c_t)
    else:
        fd = search.FlatDecoder(agent.decoder, agent.store, ctx, seq_mask)
        fd.seed(h_t, c_t)
    return env, space, fd


def _setup(agent, load_next_minibatch, key_fields=4):
    env, space, fd = _setup_space(agent, load_next_minibatch, key_fields)
    B = len(space.items)
    t = Hypotheses()
    roots = t.append(np.zeros(B, F32), np.ones(B, bool), parent=-1, inst=np.arange(B), sid=space.root_sid,
                     key=space.root_key, action=-1, count=0, pool=np.arange(B))
    return env, space, fd, t, roots


# ------------------------------------------------------------------------------------------- beam search
@_gc_paused
def beam_search(agent, beam_size, load_next_minibatch=True, mask_undo=False):
    """Seq2SeqAgent.beam_search (follower.py:541-718): per instance the `beam_size` best partial paths are
    extended by their `beam_size` best actions each step; a path is complete when it stops or reaches the
    episode length.  Returns (trajs, completed hypotheses per instance, None)."""
    env, space, fd, t, frontier = _setup(agent, load_next_minibatch)
    assert env.beam_size >= beam_size
    B = len(space.items)
    done = [[] for _ in range(B)]                       # completed hypotheses in completion order
    n_done = np.zeros(B, np.int64)
    for step in range(agent.episode_len):
        inputs, sid = _step_inputs(space, t, frontier)
        base, top_a, top_lp = fd.step_arrays(inputs, beam_size)
        N, k = top_a.shape
        owner = np.repeat(np.arange(N), k)
        act, lp = top_a.reshape(-1), top_lp.reshape(-1)
        ok = (act >= 0) & (act < inputs['a_num'][owner])                   # is_valid (follower.py:477)
        owner, act, lp = owner[ok], act[ok].astype(np.int64), lp[ok]
        par = frontier[owner]
        inst = t.inst[par]
        score = (t.score[par] + lp.astype(F32)).astype(F32)
        # per instance: best `beam_size` successors, ties in generation order (a stable descending sort)
        order = np.lexsort((np.arange(len(score)), -score, inst))
        order = order[_first_k_per_group(inst[order], beam_size)]
        owner, act, par, inst, score = owner[order], act[order], par[order], inst[order], score[order]
        nsid, stay = space.successors(sid[owner], act)
        ids = t.append(score, t.start_pose[par] & stay, parent=par, inst=inst, sid=nsid,
                       key=np.where(stay, t.key[par], space.key_of(nsid, inst)), action=act,
                       count=t.count[par] + 1, pool=base + owner)
        final = (act == 0) | (step == agent.episode_len - 1)
        for n, b in zip(ids[final], inst[final]):
            done[b].append(int(n))
        np.add.at(n_done, inst[final], 1)
        keep = ~final & (n_done[inst] < beam_size)       # an instance with enough completions stops growing
        frontier = ids[keep]
        if len(frontier) == 0:
            break
    best = []
    for lst in done:
        sc = t.score[lst]
        best.append([lst[i] for i in np.lexsort((np.arange(len(lst)), -sc))[:beam_size]])
    return _trajectories(fd, t, space, best, agent.episode_len), [HypList(t, space, lst) for lst in done], None


# --------------------------------------------------------------------------------- state-factored search
class _KeyTable:
    """One of the reference's per-instance dictionaries {world state key: (hypothesis, expanded)} for the whole
    minibatch: a dense [instance, key] array of hypotheses, and the dictionary's ENTRIES in insertion order as rows
    of [instance, slot] matrices -- `pick_score[b, slot]` = the score of the entry's current hypothesis while it
    waits to be expanded, -inf once it has been (or for a slot not in use).  "The best entry not expanded yet" of
    every instance is then one argmax over a [B, slots] matrix (first maximum = oldest entry: the tie order of the
    reference's heapq.nlargest over an insertion-ordered dict), whatever the number of entries the search has
    accumulated."""

    def __init__(self, B, n_keys, cap=128):
        self.node = np.full((B, n_keys), -1, np.int64)
        self.slot = np.full((B, n_keys), -1, np.int32)
        self.n_slots = np.zeros(B, np.int64)
        self.pick_score = np.full((B, cap), -np.inf, F32)
        self.slot_key = np.zeros((B, cap), np.int64)

    def put(self, inst, key, node, score):
        """Entries (inst, key) <- node (waiting to be expanded again).  inst grouped (equal values adjacent), in
        processing order: new entries take the instance's next slots in that order."""
        if len(inst) == 0:
            return
        slot = self.slot[inst, key]
        fresh = slot < 0
        if fresh.any():
            fi = inst[fresh]
            rank = np.arange(len(fi)) - np.flatnonzero(np.r_[True, fi[1:] != fi[:-1]])[
                np.cumsum(np.r_[True, fi[1:] != fi[:-1]]) - 1]
            new = self.n_slots[fi] + rank
            np.add.at(self.n_slots, fi, 1)
            need = int(new.max()) + 1
            if need > self.pick_score.shape[1]:
                cap = max(2 * self.pick_score.shape[1], need)
                for name, fill in (('pick_score', -np.inf), ('slot_key', 0)):
                    old = getattr(self, name)
                    grown = np.full((old.shape[0], cap), fill, old.dtype)
                    grown[:, :old.shape[1]] = old
                    setattr(self, name, grown)
            slot[fresh] = new
            self.slot[fi, key[fresh]] = new
            self.slot_key[fi, new] = key[fresh]
        self.node[inst, key] = node
        self.pick_score[inst, slot] = score


def _inputs_from_block(space, block, n):
    """The index-form inputs of `search.FlatDecoder.step_arrays` from the [8, cap] block of
    frontier_core.cpp:fill_inputs (host-issued decoder steps: fakes in the host tests, agents without a graph)."""
    h = space.h
    row, prow, view, pview, act, hrow, crow = (block[j, :n].astype(np.int64) for j in range(7))
    sid, psid = row * V + view, prow * V + pview
    return dict(vp=h['feat_row'][row], view=view, a_num=h['a_num'][sid], cand_view=h['cand_view'][sid],
                sincos=h['sincos'][sid], hrow=hrow, crow=crow, has_u=act != 0,      # (an expanded state never follows
                u_vp=h['feat_row'][prow], u_view=h['cand_view'][psid, act],         # a stop: only roots have act 0)
                u_sincos=h['sincos'][psid, act])


@_gc_paused
def state_factored_search(agent, completion_size, successor_size, load_next_minibatch=True, mask_undo=False,
                          first_n_ws_key=4):
    """Seq2SeqAgent.state_factored_search (follower.py:720-980): best-first search over WORLD STATES -- per
    instance and state only the best-scoring hypothesis is kept; every iteration expands, per instance, the
    `successor_size` best states not expanded yet (a finished hypothesis is "expanded" by recording its state
    as completed) until `completion_size` distinct end states are completed.  Returns (trajs, completed
    hypotheses, the physical traversal of every instance: the walk between successively expanded states).

    Per iteration: ONE hipGraph replay on the device (search.GraphStep) and ONE native call for the bookkeeping
    (sim/frontier_core.cpp).  `agent.search_backend = 'numpy'` (or a `tie_log`) runs the numpy restatement below
    with host-issued decoder steps instead; both produce the same hypotheses, node for node."""
    if getattr(agent, 'tie_log', None) is not None or getattr(agent, 'search_backend', 'native') == 'numpy':
        return _state_factored_search_numpy(agent, completion_size, successor_size, load_next_minibatch, mask_undo,
                                            first_n_ws_key)
    from .sim import load_frontier
    import time
    marks = getattr(agent, 'search_marks', None)         # tools/pragmatic_profile.py: wall-clock marks of the phases
    mark = (lambda name: marks.append((name, time.perf_counter()))) if marks is not None else (lambda name: None)
    mark('start')
    n_inst = getattr(agent.env, 'batch_size', None) or len(agent.env.batch)
    use_graph = getattr(agent, 'search_graph', True) and hasattr(agent.decoder, 'visual_attention_layer')
    cap = n_inst * successor_size
    env, space, fd = _setup_space(agent, load_next_minibatch, first_n_ws_key, graph_cap=cap if use_graph else 0)
    assert env.beam_size >= successor_size
    h = space.h
    core = load_frontier().StateFactored(completion_size, successor_size, agent.episode_len, first_n_ws_key, V,
                                         h['next_row'], h['cand_view'], h['a_num'], space.base_row, space.root_sid,
                                         space.root_key)
    block = fd.inputs if use_graph else np.zeros((8, cap), np.int32)
    mark('setup')
    native_loop = use_graph and getattr(agent, 'search_native_loop', True) and fd.native_loop_ready()
    while not native_loop:
        if use_graph:
            base = fd.n
            logp = fd.run(core.fill_inputs(block, base))
        else:
            n = core.fill_inputs(block, 0)
            base, logp = fd.step_logprobs(_inputs_from_block(space, block, n))
        if core.advance(logp, base) == 0 or core.done():
            break
    while native_loop:
        # the same loop inside the native module: inputs -> graph launch -> stream sync -> bookkeeping without coming
        # back to Python between iterations (sim/frontier_core.cpp: run_graph); it returns when the pool must grow
        status, fd.n, _ = core.run_graph(*fd.native_launch_args(), block, fd.out, fd.n, fd.pool_rows)
        if status == 0:
            break
        if status != 1:
            raise RuntimeError('hipGraphLaunch / hipStreamSynchronize failed in the native search loop (hip error %d)' % -status)
        fd._grow(fd.n + cap)
    mark('iterations')
    t = Hypotheses.from_arrays(*core.hypotheses())
    completed, visits = core.results()
    # (results first: a candidates_hook issues device work the rest of this function then runs beside)
    trajs = _trajectories(fd, t, space, completed, agent.episode_len, getattr(agent, 'candidates_hook', None),
                          fp32_steps=True)
    mark('results')
    traversed = _Walks(t, space, visits, agent.episode_len)
    mark('walks')
    return trajs, [HypList(t, space, lst) for lst in completed], traversed


def _state_factored_search_numpy(agent, completion_size, successor_size, load_next_minibatch=True, mask_undo=False,
                                 first_n_ws_key=4):
    """state_factored_search with the per-instance dictionaries as dense [instance, key] numpy tables updated over
    the whole minibatch: the readable restatement of what sim/frontier_core.cpp does per instance, and its
    cross-check (tests/test_search_host.py)."""
    env, space, fd, t, roots = _setup(agent, load_next_minibatch, first_n_ws_key)
    assert env.beam_size >= successor_size
    B, K = len(space.items), space.n_keys
    open_t, held_t = _KeyTable(B, K), _KeyTable(B, K)    # states to expand / finished hypotheses waiting their turn
    comp_node = np.full((B, K), -1, np.int64)            # completed end states
    n_comp = np.zeros(B, np.int64)
    comp_order = [[] for _ in range(B)]                  # keys in the order they were first completed
    # (ties between equal scores go to the older entry, states-to-expand before finished ones: heapq.nlargest over
    # chain(cache, holding), follower.py:861-865)
    all_b = np.arange(B)
    open_t.put(all_b, space.root_key.copy(), roots, np.full(B, -np.inf, F32))      # the roots: expanded from the start
    frontier = roots
    visits = [[int(r)] for r in roots]                   # successively expanded hypotheses per instance
    episode_len = agent.episode_len
    while (n_comp < completion_size).any():
        inputs, sid = _step_inputs(space, t, frontier)
        base, logp = fd.step_logprobs(inputs)
        # ---- all successors of all frontier states
        act, owner = _ragged_arange(inputs['a_num'])
        par = frontier[owner]
        inst = t.inst[par]
        live = n_comp[inst] < completion_size
        act, owner, par, inst = act[live], owner[live], par[live], inst[live]
        score = (t.score[par] + logp[owner, act]).astype(F32)
        count = t.count[par] + 1
        nsid, stay = space.successors(sid[owner], act)
        key = np.where(stay, t.key[par], space.key_of(nsid, inst))
        final = (act == 0) | (count == episode_len)
        # the reference walks each instance's successors in descending score order (stable) and keeps, per
        # state, a successor only if it beats what the table holds: the first of every (instance, table, key)
        # group in that order is the only one that can win
        order = np.lexsort((np.arange(len(score)), -score, inst))
        group = (inst[order] * 2 + final[order]) * K + key[order]
        _, first = np.unique(group, return_index=True)
        cand = order[np.sort(first)]                     # winners-to-be, in processing order
        ci, ck, cf = inst[cand], key[cand], final[cand]
        cur = np.where(cf, held_t.node[ci, ck], open_t.node[ci, ck])
        wins = (cur < 0) | (t.score[np.maximum(cur, 0)] < score[cand])
        cand, ci, ck, cf, cur = cand[wins], ci[wins], ck[wins], cf[wins], cur[wins]
        ids = t.append(score[cand], t.start_pose[par[cand]] & stay[cand], parent=par[cand], inst=ci, sid=nsid[cand],
                       key=ck, action=act[cand], count=count[cand], pool=base + owner[cand])
        for tab, m in ((held_t, cf), (open_t, ~cf)):
            tab.put(ci[m], ck[m], ids[m], score[cand[m]])
        # ---- per instance: the `successor_size` best entries not expanded yet
        active = np.flatnonzero(n_comp < completion_size)
        tie_log = getattr(agent, 'tie_log', None)
        picks = []
        for rnd in range(successor_size):
            so, sh = open_t.pick_score[active], held_t.pick_score[active]
            ao, ah = so.argmax(1), sh.argmax(1)
            r = np.arange(len(active))
            mo, mh = so[r, ao], sh[r, ah]
            use_h = mh > mo                                   # equal scores: the state to expand first
            got = np.maximum(mo, mh) > -np.inf
            if tie_log is not None and successor_size == 1:
                # diagnostics (tools/search_tie_probe.py): per pick, how far behind the runner-up of the same instance was
                best = np.where(use_h, mh, mo)
                so2, sh2 = so.copy(), sh.copy()
                so2[r[~use_h], ao[~use_h]] = -np.inf
                sh2[r[use_h], ah[use_h]] = -np.inf
                second = np.maximum(so2.max(1), sh2.max(1))
                has2 = got & (second > -np.inf)
                tie_log.append((active[has2], best[has2], second[has2]))
            a_i, u_h = active[got], use_h[got]
            slot = np.where(u_h, ah[got], ao[got])
            held_t.pick_score[a_i[u_h], slot[u_h]] = -np.inf  # expanded
            open_t.pick_score[a_i[~u_h], slot[~u_h]] = -np.inf
            key_ = np.empty(len(a_i), np.int64)
            key_[u_h] = held_t.slot_key[a_i[u_h], slot[u_h]]
            key_[~u_h] = open_t.slot_key[a_i[~u_h], slot[~u_h]]
            picks.append((a_i, key_, u_h))
            if not got.all():
                active = active[got]                          # (an instance with nothing left has nothing later either)
        pi, pk, ph = (np.concatenate(x) for x in zip(*picks))
        if successor_size > 1:
            o = np.argsort(pi, kind='stable')                 # instance-major, each instance's picks best first
            pi, pk, ph = pi[o], pk[o], ph[o]
        pn = np.where(ph, held_t.node[pi, pk], open_t.node[pi, pk])
        # finished hypotheses: their end state is completed (the better one if it already was)
        fi, fk, fn = pi[ph], pk[ph], pn[ph]
        old = comp_node[fi, fk]
        better = (old < 0) | (t.score[np.maximum(old, 0)] < t.score[fn])
        comp_node[fi[better], fk[better]] = fn[better]
        for b, k_ in zip(fi[old < 0], fk[old < 0]):
            comp_order[b].append(int(k_))
        np.add.at(n_comp, fi[old < 0], 1)
        # states to expand next; an instance that has reached its completions stops
        grow = ~ph & (n_comp[pi] < completion_size)
        frontier = pn[grow]
        if len(frontier) == 0:
            break
        for b, n in zip(pi[grow], frontier):
            visits[b].append(int(n))
    completed = []
    for b in range(B):
        nodes = [int(comp_node[b, k_]) for k_ in comp_order[b]]
        sc = t.score[nodes]
        completed.append([nodes[i] for i in np.lexsort((np.arange(len(nodes)), -sc))[:completion_size]])
        visits[b].extend(completed[b])
    traversed = [HypList(t, space, w) for w in physical_walks(t, visits, episode_len)]
    return (_trajectories(fd, t, space, completed, episode_len, fp32_steps=True),
            [HypList(t, space, lst) for lst in completed], traversed)


class _Walks(Sequence):
    """The physical traversal of every instance (a list of HypList), worked out when first read: only
    `physical_traversal=True` (rational_follower.py:87-96) looks at it."""

    def __init__(self, t, space, visits, depth):
        self._args, self._lists = (t, space, visits, depth), None

    def _get(self):
        if self._lists is None:
            t, space, visits, depth = self._args
            self._lists = [HypList(t, space, w) for w in physical_walks(t, visits, depth)]
        return self._lists

    def __len__(self):
        return len(self._args[2])

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())


def physical_walks(t, visits, depth):
    """The walk an agent would really make between successively expanded hypotheses (follower.py:52-73,
    768-781), for every instance at once: from each hypothesis up its lineage to the nearest ancestor standing
    on a viewpoint the next one's lineage also visits, then down that lineage from its OLDEST state on that
    viewpoint.  visits: per instance the expanded hypotheses in order.  Returns per instance the node ids."""
    a = np.array([n for seq in visits for n in seq[:-1]], np.int64)
    b = np.array([n for seq in visits for n in seq[1:]], np.int64)
    if len(a) == 0:
        return [list(seq) for seq in visits]
    LA, _ = _lineage_matrix(t, a, depth)
    LB, _ = _lineage_matrix(t, b, depth)
    RA = np.where(LA >= 0, t.sid[np.maximum(LA, 0)] // V, -1)       # viewpoint (nav row) of every ancestor
    RB = np.where(LB >= 0, t.sid[np.maximum(LB, 0)] // V, -2)
    M = RA[:, :, None] == RB[:, None, :]                            # [pairs, up index, down index]
    assert M.any(axis=(1, 2)).all(), 'two hypotheses of one instance share no viewpoint'
    ia = M.any(2).argmax(1)                                         # first ancestor of a on a shared viewpoint
    Mi = M[np.arange(len(a)), ia]                                   # [pairs, down index]
    ib = Mi.shape[1] - 1 - Mi[:, ::-1].argmax(1)                    # the OLDEST state of b's lineage standing there
    # pair p contributes LA[p, 1 .. ia] (walking back) then LB[p, ib-1 .. 0] (walking forward to b)
    off, owner = _ragged_arange(ia + ib)
    back = off < ia[owner]
    last = LA.shape[1] - 1                                          # (both branches are evaluated: clamp the unused one)
    steps = np.where(back, LA[owner, np.minimum(off + 1, last)],
                     LB[owner, np.clip(ib[owner] - 1 - (off - ia[owner]), 0, last)])
    per_pair = np.cumsum(ia + ib)
    walks, p0 = [], 0
    for seq in visits:
        n_pairs = len(seq) - 1
        lo = per_pair[p0 - 1] if p0 else 0
        hi = per_pair[p0 + n_pairs - 1] if n_pairs else lo
        walks.append([seq[0]] + steps[lo:hi].tolist())
        p0 += n_pairs
    return walks


# ------------------------------------------------------------------------------------ speaker beam search
@_gc_paused
def speaker_beam_search(speaker, beam_size, path_obs, path_actions):
    """Seq2SeqSpeaker.beam_search (speaker.py:211-318): one flat SpeakerDecoderLSTM step per word over all live
    hypotheses of all paths; hypotheses are rows (parent, path, word, float32 score, pool row)."""
    from . import search
    start_obs, feats, acts, path_mask, _, _, perm = speaker._batch_observations_and_actions(path_obs, path_actions, None)
    B = len(start_obs)
    with torch.no_grad():
        ctx, h_t, c_t = speaker.encoder(acts, feats)
    sd = search.FlatSpeakerDecoder(speaker.decoder, ctx.detach(), path_mask)
    sd.seed(h_t.detach(), c_t.detach())
    parent, inst, word, pool = [np.full(B, -1)], [np.arange(B)], [np.full(B, search.BOS)], [np.arange(B)]
    score = [np.zeros(B, F32)]
    n_nodes = B
    frontier = np.arange(B)
    done = [[] for _ in range(B)]
    n_done = np.zeros(B, np.int64)
    P, I, W, R, S = (np.concatenate(x) for x in (parent, inst, word, pool, score))
    for step in range(speaker.instruction_len):
        base, top_w, top_lp = sd.step(W[frontier], R[frontier], I[frontier], beam_size)
        N, k = top_w.shape
        owner = np.repeat(np.arange(N), k)
        par = frontier[owner]
        sc = (S[par] + top_lp.reshape(-1).astype(F32)).astype(F32)
        order = np.lexsort((np.arange(len(sc)), -sc, I[par]))
        order = order[_first_k_per_group(I[par][order], beam_size)]
        owner, par, sc, wd = owner[order], par[order], sc[order], top_w.reshape(-1)[order].astype(np.int64)
        ids = np.arange(n_nodes, n_nodes + len(sc))
        n_nodes += len(sc)
        P, I, W, R, S = (np.concatenate((P, par)), np.concatenate((I, I[par])), np.concatenate((W, wd)),
                         np.concatenate((R, base + owner)), np.concatenate((S, sc)))
        final = (wd == search.EOS) | (step == speaker.instruction_len - 1)
        for n, b in zip(ids[final], I[ids[final]]):
            done[b].append(int(n))
        np.add.at(n_done, I[ids[final]], 1)
        frontier = ids[~final & (n_done[I[ids]] < beam_size)]
        if len(frontier) == 0:
            break
    tok = getattr(speaker.env, 'tokenizer', None)
    outputs = [[] for _ in range(B)]
    for b, src in enumerate(perm):
        assert not outputs[src]
        lst = done[b]
        for i in np.lexsort((np.arange(len(lst)), -S[lst]))[:beam_size]:
            lin = []
            n = lst[i]
            while n >= 0:
                lin.append(n)
                n = P[n]
            lin = lin[::-1]                              # BOS root first
            sc = [float(S[n]) for n in lin]
            words = [int(W[n]) for n in lin[1:]]
            outputs[src].append({
                'instr_id': start_obs[b]['instr_id'], 'word_indices': words, 'score': sc[-1],
                'scores': [y - x for x, y in zip(sc, sc[1:])],
                'words': tok.decode_sentence(words, break_on_eos=True, join=False) if tok is not None else list(words),
                'attentions': sd.attention_rows([int(R[n]) for n in lin[1:]])})
    return outputs
