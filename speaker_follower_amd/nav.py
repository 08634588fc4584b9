"""Device-resident navigation (BASELINE configs[1]: student-forced rollouts on real
R2R items without a host round trip per step).

The reference walks the simulator from Python every step (`R2RBatch.step` / `observe`,
tasks/R2R/env.py:628-641, 763-804; the 36-view sweep :149-224; the teacher :742-761) and syncs the
chosen actions to the host to do so (follower.py:507-514).  Everything that walk computes is a pure
function of the state (scan, viewpoint, view index):

  * `NavTable`        -- built ONCE per set of connectivity graphs with this repo's navigation-only
                         simulator: per state the candidate list (next viewpoint, absViewIndex, relative
                         angles) exactly as `env.panorama_sweep` orders it, over contiguous "nav rows";
  * `DeviceNavBatch`  -- one R2R minibatch: instructions, start states, and per sample the next hop of
                         the shortest path to ITS goal from every viewpoint of its scan (the teacher);
  * `sf_nav_step`     -- the per-step kernel (csrc/sf_nav.hip): state + a_t -> next state, written as
                         the next decode step's index-form observation.

`FollowerEngine.rollout(DeviceNavBatch, ...)` then runs encoder + S decode steps with ONE host sync at
the end; `trajectories()` turns the recorded states into the reference's result format.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from . import sim as _sim
from ._lib import call
from .env import ANGLE_INC, IMAGE_H, IMAGE_W, VFOV, WorldState
from .lazydict import LazyDict
from .features import cand_sincos
from .follower import batch_instructions_from_encoded
from .runtime import ptr, stream

V = 36


class NavTable:
    """Candidate lists of every (viewpoint, view) state of the env's graphs, on the device."""

    def __init__(self, env, store, a_max=None):
        """Rows = the included viewpoints of the env's graphs, scan by scan in connectivity-file order.  The
        candidate lists come from the batched native sweep (sim/sweep_py.cpp: one call per scan, bit-identical
        to env.panorama_sweep); a sweep does not depend on the agent's elevation, so each of the 12 heading
        tables serves three views."""
        sweep = _sim.load_sweep().sweep_scan
        self.scans = sorted(env.graphs)
        self.row_of = {}                 # (scan, viewpoint) -> nav row
        self.vp_of = []                  # nav row -> (scan, viewpoint)
        self.base = {}                   # scan -> first nav row
        self.scan_rows = {}
        per_scan = []
        A = max(1, a_max or 0)
        for s in self.scans:
            g = env.graphs[s]
            nodes = [v for v, inc in zip(g.ids, g.included) if inc]
            self.base[s] = len(self.vp_of)
            self.scan_rows[s] = len(nodes)
            for v in nodes:
                self.row_of[(s, v)] = len(self.vp_of)
                self.vp_of.append((s, v))
            tabs = sweep(env.nav_graph_path, s, nodes, IMAGE_W, IMAGE_H, math.radians(VFOV), 0)
            per_scan.append(tabs)
            A = max(A, tabs[1].shape[2])
        self.A = A
        n = len(self.vp_of)
        a_num = np.zeros((n, 3, 12), np.int32)
        next_row = np.zeros((n, 3, 12, A), np.int32)
        cand_view = np.zeros((n, 3, 12, A), np.int32)
        head = np.zeros((n, 3, 12, A), np.float64)
        elev = np.zeros((n, 3, 12, A), np.float64)
        dist12 = np.zeros((n, 12, A), np.float64)              # (per heading bin: the sweep ignores the elevation)
        for s, (an, nx, av, rh, re, ds) in zip(self.scans, per_scan):
            b, m, a = self.base[s], an.shape[0], nx.shape[2]
            assert (nx >= 0).all(), 'a candidate of scan %s is not an included viewpoint' % s
            av = av.copy()
            av[:, :, 0] = 0                                    # the stop slot carries no view
            a_num[b:b + m] = an[:, None, :]
            next_row[b:b + m, :, :, :a] = (nx + b)[:, None]
            next_row[b:b + m, :, :, a:] = np.arange(b, b + m, dtype=np.int32)[:, None, None, None]
            cand_view[b:b + m, :, :, :a] = av[:, None]
            head[b:b + m, :, :, :a] = rh[:, None]
            elev[b:b + m, :, :, :a] = re[:, None]
            dist12[b:b + m, :, :a] = ds
        a_num = a_num.reshape(n * V)
        next_row, cand_view = next_row.reshape(n * V, A), cand_view.reshape(n * V, A)
        head, elev = head.reshape(n * V, A), elev.reshape(n * V, A)
        feat_row = np.array([env.row_of[s + '_' + v] for s, v in self.vp_of], np.int32)
        dev = store.device
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)   # noqa: E731
        self.a_num, self.next_row, self.cand_view = up(a_num, torch.int32), up(next_row, torch.int32), up(cand_view, torch.int32)
        sincos = cand_sincos(head, elev).astype(np.float32)               # the same host sin/cos the env batches use
        self.sincos = up(sincos, torch.float32)
        self.feat_row = up(feat_row, torch.int32)
        # host copies: the search procedures (search.py) expand states by integer table look-ups
        self.host = dict(a_num=a_num, next_row=next_row, cand_view=cand_view, sincos=sincos, feat_row=feat_row,
                         heading=head, elevation=elev, distance12=dist12)
        self.n_rows, self.device, self.env = n, dev, env
        self._hops = {}

    def adj_loc_list(self, sid):
        """The candidate list of state `sid` (= nav row * 36 + view) in the env's dictionary form (env.py:149-224 /
        env.panorama_sweep: stop first, then by |rel_heading|) straight from the tables -- equal to what the
        simulator sweep of that state returns, without waking the simulator."""
        h = self.host
        row, view = int(sid) // V, int(sid) % V
        n = int(h['a_num'][sid])
        nxt, cv = h['next_row'][sid, :n].tolist(), h['cand_view'][sid, :n].tolist()
        rh, re = h['heading'][sid, :n].tolist(), h['elevation'][sid, :n].tolist()
        ds = h['distance12'][row, view % 12, :n].tolist()
        vp_of = self.vp_of
        adj = [dict(absViewIndex=-1, nextViewpointId=vp_of[row][1], rel_heading=0.0, rel_elevation=0.0, distance=0.0)]
        for a in range(1, n):
            adj.append(dict(absViewIndex=cv[a], nextViewpointId=vp_of[nxt[a]][1], rel_heading=rh[a],
                            rel_elevation=re[a], distance=ds[a]))
        return adj

    def struct(self):
        return _lib.NavTableS(self.a_num.data_ptr(), self.next_row.data_ptr(), self.cand_view.data_ptr(),
                              self.sincos.data_ptr(), self.feat_row.data_ptr(), self.A, V)

    def gold_routes(self, items, max_steps):
        """The shortest-path routes of a minibatch (env.py:823-848: from every item's start pose follow the teacher
        action until it says stop, at most max_steps actions) in INDEX FORM, from the tables alone: n [B] int32 actions
        per route and rows [sum n, 6] float64, route-major, one per (state, action): (feature row, view index, the
        action's view index, rel_heading, rel_elevation, is_stop) -- what the speaker's scoring consumes (speaker.py:
        68-121).  Teacher and transition rules are sf_nav_step's (csrc/sf_glue.h: nav_advance_slot): stop at the goal,
        else the first candidate that leads to the next hop; a candidate that is the current viewpoint leaves the state."""
        h, env = self.host, self.env
        B = len(items)
        row = np.array([self.row_of[(it['scan'], it['path'][0])] for it in items], np.int64)
        view = np.array([env.start_view(WorldState(it['scan'], it['path'][0], it['heading'], 0)) for it in items], np.int64)
        base = np.array([self.base[it['scan']] for it in items], np.int64)
        hop_of = [self.hops(it['scan'], it['path'][-1]) for it in items]
        hop = np.zeros((B, max(len(x) for x in hop_of)), np.int64)
        for b, x in enumerate(hop_of):
            hop[b, :len(x)] = x
        A = h['next_row'].shape[1]
        cols = np.arange(A)[None, :]
        every = np.arange(B)
        alive = np.ones(B, bool)
        recs = []                                            # per step: (rows alive, their 6 columns)
        for _ in range(max_steps):
            s = row * V + view
            goal_hop = hop[every, row - base]
            nxt = h['next_row'][s]                           # [B, A]
            leads = (nxt == goal_hop[:, None]) & (cols >= 1) & (cols < h['a_num'][s][:, None])
            a = np.where((goal_hop != row) & leads.any(1), leads.argmax(1), 0)
            move = a > 0
            idx = np.flatnonzero(alive)
            sa, aa, mv = s[idx], a[idx], move[idx]
            recs.append((idx, np.stack((h['feat_row'][row[idx]], view[idx], np.where(mv, h['cand_view'][sa, aa], 0),
                                        np.where(mv, h['heading'][sa, aa], 0.0), np.where(mv, h['elevation'][sa, aa], 0.0),
                                        (~mv).astype(np.float64)), axis=1).astype(np.float64)))
            alive = alive & move
            if not alive.any():
                break
            nr = nxt[every, a]
            go = move & (nr != row)
            view = np.where(go, h['cand_view'][s, a], view)
            row = np.where(go, nr, row)
        n = np.zeros(B, np.int32)
        for idx, _ in recs:
            n[idx] += 1
        first = np.concatenate(([0], np.cumsum(n)))
        rows = np.zeros((int(first[-1]), 6), np.float64)
        for t, (idx, part) in enumerate(recs):
            rows[first[idx] + t] = part                      # (a route's steps are consecutive from step 0)
        return n, rows

    def hops(self, scan, goal):
        """[rows of `scan`] int32: next nav row on the shortest path to `goal` (itself at the goal, and
        where no path exists), the table behind env.py:742-761."""
        key = (scan, goal)
        if key not in self._hops:
            g = self.env.graphs[scan]
            b = self.base[scan]
            out = np.arange(b, b + self.scan_rows[scan], dtype=np.int32)
            nxt = g.next_hops(goal)                      # (one look at the scan's all-pairs table, env.NavGraph)
            for i in range(self.scan_rows[scan]):
                hop = nxt.get(self.vp_of[b + i][1])
                if hop is not None:
                    out[i] = self.row_of[(scan, hop)]
            self._hops[key] = out
        return self._hops[key]


class DeviceNavBatch:
    """The current minibatch of an R2RIndexEnv as device state for `FollowerEngine.rollout`:
    `vp/view/a_num/cand_view/sincos/target` are [S+1, B, ...] buffers that `advance` fills one step
    ahead of the decoder (slot 0 = the initial observation)."""

    def __init__(self, nav, items, steps, max_length=80, reverse=True, row0=0, fixed_shapes=False, host=None):
        """fixed_shapes: every tensor gets the shape of the LARGEST minibatch of this size (instructions padded to
        max_length, goal-hop rows as long as the largest scan) so that `load(items)` can refresh the batch in place --
        what a captured training graph reads (runtime.TrainingGraph) must keep its addresses."""
        dev = nav.device
        self.nav, self.items, self.steps = nav, items, steps
        self.max_length, self.reverse, self.fixed = max_length, reverse, fixed_shapes
        B, A, S = len(items), nav.A, steps
        h = host if host is not None else self._host_arrays(items)       # (host: host_arrays_for(...) of these items)
        self.lengths = h['lengths'].tolist()
        to = lambda x: torch.from_numpy(x).to(dev)                                        # noqa: E731
        self.seq, self.mask, self.lengths_dev = to(h['seq']), to(h['mask']), to(h['lengths'])
        self.a_max, self.row0 = A, row0
        z = lambda *s, dt=torch.int32: torch.zeros(*s, dtype=dt, device=dev)             # noqa: E731
        self.row = z(S + 1, B)
        self.vp, self.view, self.a_num = z(S + 1, B), z(S + 1, B), z(S + 1, B)
        self.cand_view = z(S + 1, B, A)
        self.sincos = z(S + 1, B, A, 4, dt=torch.float32)
        self.target = z(S + 1, B, dt=torch.int64)
        self.row0_state, self.view0_state = to(h['rows']), to(h['views'])
        self.goal_hop, self.hop_base = to(h['hop']), to(h['base'])
        self.ld_hop = h['hop'].shape[1]
        self._nav_struct = nav.struct()
        if fixed_shapes and dev.type == 'cuda':
            self._pack_for_load()

    _LOADED = (('seq', 'seq'), ('mask', 'mask'), ('lengths_dev', 'lengths'), ('row0_state', 'rows'), ('view0_state', 'views'),
               ('goal_hop', 'hop'), ('hop_base', 'base'))

    def _pack_for_load(self):
        """The seven tensors `load` rewrites become views of ONE device buffer with a pinned host mirror: a minibatch is
        one asynchronous H2D copy instead of seven pageable ones (0.2 ms of the training loop's host time)."""
        offs, total = {}, 0
        for attr, _ in self._LOADED:
            t = getattr(self, attr)
            offs[attr] = total
            total += (t.numel() * t.element_size() + 63) & ~63
        dev_buf = torch.empty(total, dtype=torch.uint8, device=self.seq.device)
        self._pack_host = torch.empty(total, dtype=torch.uint8).pin_memory()
        self._pack_np = {}
        for attr, key in self._LOADED:
            t = getattr(self, attr)
            n = t.numel() * t.element_size()
            view = dev_buf[offs[attr]:offs[attr] + n].view(t.dtype).view(t.shape)
            view.copy_(t)
            setattr(self, attr, view)
            self._pack_np[key] = self._pack_host[offs[attr]:offs[attr] + n].view(t.dtype).view(t.shape).numpy()
        self._pack_dev = dev_buf

    def _host_arrays(self, items):
        return self.host_arrays_for(self.nav, items, self.max_length, self.reverse, self.fixed)

    @staticmethod
    def host_arrays_for(nav, items, max_length=80, reverse=True, fixed=False):
        """What a minibatch contributes: the encoded instructions (follower.py:75-105) and, per item, the start state
        (newEpisode snaps the item's heading to the discrete view, env.py:814-819) and the hop table towards its goal.
        Host work only: callers form it for the NEXT minibatch while the device runs the current one."""
        env = nav.env
        seq, mask, lengths = batch_instructions_from_encoded([it['instr_encoding'] for it in items], max_length,
                                                             reverse=reverse, device='cpu')
        seq = seq.numpy()
        mask = (seq == 0) if fixed else mask.numpy()                                      # (PAD = 0; full width when fixed)
        B = len(items)
        ld = max(nav.scan_rows.values()) if fixed else max(nav.scan_rows[it['scan']] for it in items)
        rows, views = np.zeros(B, np.int32), np.zeros(B, np.int32)
        hop, base = np.zeros((B, ld), np.int32), np.zeros(B, np.int32)
        for b, it in enumerate(items):
            views[b] = env.start_view(WorldState(it['scan'], it['path'][0], it['heading'], 0))
            rows[b] = nav.row_of[(it['scan'], it['path'][0])]
            hp = nav.hops(it['scan'], it['path'][-1])
            hop[b, :len(hp)] = hp
            base[b] = nav.base[it['scan']]
        return dict(seq=np.ascontiguousarray(seq), mask=np.ascontiguousarray(mask.astype(np.uint8)),
                    lengths=np.asarray(lengths, np.int32), rows=rows, views=views, hop=hop, base=base)

    def load(self, items, host=None):
        """The next minibatch INTO the same device tensors (fixed_shapes only): seven small H2D copies on the current
        stream; the observation slots are rewritten by the rollout itself.  `host`: what `_host_arrays(items)` returned
        earlier (the agents' training loop forms it while the device still runs the previous iteration)."""
        if not self.fixed:
            raise ValueError('DeviceNavBatch.load needs fixed_shapes=True')
        if len(items) != self.batch_size:
            raise ValueError('DeviceNavBatch.load: %d items for a batch of %d' % (len(items), self.batch_size))
        h = host if host is not None else self._host_arrays(items)
        self.items, self.lengths = items, h['lengths'].tolist()
        if getattr(self, '_pack_dev', None) is not None:
            # ONE pinned mirror, rewritten per minibatch: the previous asynchronous copy OUT of it must have completed
            # before the host writes it again.  Nothing in the pipelined training loop guarantees that by itself (the
            # copy is queued behind the previous replay; the host runs ahead of the device), so the copy is followed
            # by an event and the next load waits on it -- normally long reached, then the wait costs nothing.
            ev = getattr(self, '_pack_copied', None)
            if ev is not None:
                ev.synchronize()
            for key, dst in self._pack_np.items():
                np.copyto(dst, h[key], casting='same_kind')
            self._pack_dev.copy_(self._pack_host, non_blocking=True)
            if ev is None:
                ev = self._pack_copied = torch.cuda.Event()
            ev.record()
            return
        for attr, key in self._LOADED:
            getattr(self, attr).copy_(torch.from_numpy(h[key]))

    @property
    def batch_size(self):
        return self.seq.shape[0]

    def advance(self, t, a_t=None, ended=None):
        """Fills slot t + 1 from slot t and the actions of step t (t = -1: the initial observation)."""
        src_row = self.row0_state if t < 0 else self.row[t]
        src_view = self.view0_state if t < 0 else self.view[t]
        n = t + 1
        call('sf_nav_step', C.byref(self._nav_struct), self.batch_size, ptr(src_row), ptr(src_view),
             ptr(a_t) if a_t is not None else None, ptr(ended) if ended is not None else None,
             ptr(self.goal_hop), self.ld_hop, ptr(self.hop_base), ptr(self.row[n]), ptr(self.vp[n]),
             ptr(self.view[n]), ptr(self.a_num[n]), ptr(self.cand_view[n]), ptr(self.sincos[n]),
             ptr(self.target[n]), stream())

    def fused_step(self, t):
        """sf_nav_io of decode step t (slot t -> slot t + 1) for sf_follower_glue.nav: the env step runs in
        the scoring + glue launch, right behind the action choice, instead of `advance(t, ...)`."""
        n = t + 1
        return _lib.NavIO(self._nav_struct, self.row[t].data_ptr(), self.view[t].data_ptr(),
                          self.goal_hop.data_ptr(), self.ld_hop, self.hop_base.data_ptr(),
                          self.row[n].data_ptr(), self.vp[n].data_ptr(), self.view[n].data_ptr(),
                          self.a_num[n].data_ptr(), self.cand_view[n].data_ptr(), self.sincos[n].data_ptr(),
                          self.target[n].data_ptr())

    def trajectories(self, st):
        """The rollout's result dictionaries (follower.py:446-456, 517-524): per sample instr_id,
        trajectory [(viewpointId, heading, elevation)], actions, scores -- the stop action and the
        duplicated final state included, nothing after it.  The device arrays come down once; the per-sample lists are
        cut from them with numpy, and 'trajectory' (a tuple per visited pose) is built when somebody reads it."""
        S = st.steps
        return self.trajectories_from(self.items, S, self.row[:S + 1].cpu().numpy(), self.view[:S + 1].cpu().numpy(),
                                      st.actions.cpu().numpy(), st.step_scores.cpu().numpy())

    def trajectories_from(self, items, S, rows, views, acts, sc):
        """`trajectories` over host copies of the rollout's arrays (rows / views [S+1,B], actions / step scores [S,B]); the
        result dictionaries keep slices of them."""
        B = len(items)
        stopped = acts[:S] == 0
        n = np.where(stopped.any(0), stopped.argmax(0) + 1, S)                  # steps up to and including the stop action
        totals = np.cumsum(sc[:S], axis=0, dtype=np.float32)                    # (sequential float32 sums, as the loop's)
        acts_t, sc_t = np.ascontiguousarray(acts[:S].T), np.ascontiguousarray(sc[:S].T)
        last = totals[n - 1, np.arange(B)].tolist()
        return [_Trajectory(self.nav, it, int(n[b]), acts_t[b], sc_t[b], last[b], rows[:, b], views[:, b])
                for b, it in enumerate(items)]


class _Trajectory(LazyDict):
    """One result of a device rollout: 'instr_id', 'actions', 'scores', 'score' are there, 'trajectory' is made on demand."""
    __slots__ = ('_nav', '_it', '_n', '_rows', '_views')

    def __init__(self, nav, it, n, acts, sc, score, rows, views):
        LazyDict.__init__(self, {'instr_id': it['instr_id'], 'actions': acts[:n].tolist(), 'scores': sc[:n].tolist(),
                                 'score': score}, ('trajectory',))
        self._nav, self._it, self._n, self._rows, self._views = nav, it, n, rows, views

    def _make(self, key):
        # (vp, heading, elevation) of every observation on the way: the SNAPPED pose of the simulator state (env.py:783-784),
        # for the start pose too -- an item's continuous start heading never appears in a result
        vp_of = self._nav.vp_of
        out = []
        for t in range(self._n + 1):
            v = int(self._views[t])
            out.append((vp_of[self._rows[t]][1], (v % 12) * ANGLE_INC, (v // 12 - 1) * ANGLE_INC))
        return out


def table_for(env, store):
    """The NavTable of `env`'s graphs over `store`, built once per (env, store) pair."""
    cached = getattr(env, '_nav_table', None)
    if cached is None or cached[0] is not store or cached[2] != tuple(sorted(env.graphs)):
        from .build import build_sim
        build_sim(verbose=False)
        env._nav_table = (store, NavTable(env, store), tuple(sorted(env.graphs)))
    return env._nav_table[1]
