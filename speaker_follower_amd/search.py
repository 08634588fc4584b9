"""Search procedures of the follower and the speaker on the HIP path:

* `beam_search`            -- Seq2SeqAgent.beam_search            (follower.py:541-718)
* `state_factored_search`  -- Seq2SeqAgent.state_factored_search  (follower.py:720-980)
* `speaker_beam_search`    -- Seq2SeqSpeaker.beam_search          (speaker.py:211-318)
* `rational_mix`, `run_rational_follower` -- the pragmatic re-ranking (rational_follower.py:12-190)

The three searches live in frontier.py: hypotheses are rows of integer / float32 arrays with parent pointers,
world states are integers into the navigation tables (nav.NavTable), the per-instance bookkeeping is numpy over
the whole minibatch.  This module holds what they share with the rest of the package: the flat decoder steps
on the device, the helpers the reference exports from follower.py (`least_common_viewpoint_path`,
`backchain_inference_states`, `InferenceState`) and the re-ranking.  Every iteration is ONE flat decoder
step over all live states on the device:

  reference (per iteration)                            here
  -----------------------------------------------     ------------------------------------------------
  np.stack of N x 36 x 2176 features + H2D             N x (row, view) int32 -> kernels gather from HBM
  all_u_t [N,A,2176] built on the host + H2D           N x A (view, sin/cos) -> gathered in the kernel
  h_t[flat_indices], c_t[flat_indices]                 sf_gather_rows from the state pool
  ctx[beam_indices], seq_mask[beam_indices] (copies)   ctx_row indirection inside the attention kernel
  log_softmax + topk + gather, .data on the host       sf_logprob_topk, one D2H of [N,k] per iteration
  last_action_embedding = all_u_t[i, a] (2176 floats)  (row, view, sin/cos) of the parent's table entry
  env.step / env.observe per successor (Python)        next_row[s, a], cand_view[s, a] look-ups (numpy)

The agent needs a `features.FeatureStore` (`agent.store`) and an env.R2RIndexEnv.
"""
import ctypes as C
from collections import namedtuple, Counter

import numpy as np
import torch

from . import _lib
from ._lib import call
from .features import cand_sincos
from .follower import batch_instructions_from_encoded, EOS, BOS
from .model import decoder_params, decoder_w_struct, decoder_tape, tape_struct
from .runtime import ptr, stream, ws_args, gc_paused, graph_capture

byref = C.byref

# follower.py:19 -- h_t / c_t / last_alpha hold rows of the device state pool instead of tensors,
# last_action_embedding holds the index-form descriptor of the embedding instead of 2176 floats
InferenceState = namedtuple(
    'InferenceState',
    'prev_inference_state, world_state, observation, flat_index, last_action, '
    'last_action_embedding, action_count, score, h_t, c_t, last_alpha')

SpeakerInferenceState = namedtuple(
    'SpeakerInferenceState',
    'prev_inference_state, flat_index, last_word, word_count, score, last_alpha')   # speaker.py:18

def _add_f32(score, step_score):
    """The reference accumulates hypothesis scores as `score + action_score` with action_score a
    float32 tensor element (follower.py:629, 826; speaker.py:277): every partial sum is rounded to
    float32.  Same rounding here, so that ties and orderings agree."""
    return float(np.float32(score) + np.float32(step_score))


def flatten(lol):
    return [x for l in lol for x in l]                                               # utils.py:205


def path_element_from_observation(ob):
    return (ob['viewpoint'], ob['heading'], ob['elevation'])


def _lineage(inf_state):
    """The state and its ancestors, newest first (back-pointer chain of InferenceState)."""
    chain = []
    while inf_state is not None:
        chain.append(inf_state)
        inf_state = inf_state.prev_inference_state
    return chain


def backchain_inference_states(last_inference_state):
    """What follower.py:33-50 returns for a hypothesis: (world states, observations, actions,
    per-action scores, attention rows), oldest first; the start pseudo-action (and its attention
    row) is left out, and the score of action i is the difference of the cumulative scores on
    either side of it."""
    chain = _lineage(last_inference_state)[::-1]
    taken = chain[1:]
    return ([s.world_state for s in chain], [s.observation for s in chain],
            [s.last_action for s in taken],
            [s.score - p.score for p, s in zip(chain, taken)],
            [s.last_alpha for s in taken])


def least_common_viewpoint_path(inf_state_a, inf_state_b):
    """The physical walk between two frontier states (follower.py:52-73): up from A to its nearest
    ancestor standing on a viewpoint that B's lineage also visits, then down B's lineage from the
    OLDEST state on that viewpoint to B.  The meeting viewpoint appears once."""
    down = _lineage(inf_state_b)                       # B, parent(B), ..., root
    oldest_at = {s.world_state.viewpointId: i for i, s in enumerate(down)}   # later (older) index wins
    up = []
    for s in _lineage(inf_state_a):
        up.append(s)
        i = oldest_at.get(s.world_state.viewpointId)
        if i is not None:
            return up + down[:i][::-1]
    raise AssertionError('the two states share no viewpoint on their lineages')


class _Pool:
    """Rows of per-state device data that outlive the iteration that produced them (hidden and
    cell state, text attention).  Decoder steps write straight into the next free rows."""

    def __init__(self, width, device, cap=256):
        self.buf = torch.empty(cap, width, device=device, dtype=torch.float32)
        self.n = 0

    def reserve(self, n):
        if self.n + n > self.buf.shape[0]:
            cap = max(2 * self.buf.shape[0], self.n + n)
            new = torch.empty(cap, self.buf.shape[1], device=self.buf.device, dtype=torch.float32)
            new[:self.n].copy_(self.buf[:self.n])
            self.buf = new
        base = self.n
        self.n += n
        return base, self.buf[base:base + n]


class FlatDecoder:
    """One follower decoder step over a flat list of search states (a6 without the glue)."""

    def __init__(self, decoder, store, ctx, mask):
        self.dec, self.store = decoder, store
        self.ctx, self.mask = ctx.contiguous(), mask.to(torch.uint8).contiguous()
        self.dev = store.device
        self.H = decoder.hidden_size
        self.D = decoder.visual_attention_layer.linear_in_h.weight.shape[0]
        self.T = ctx.shape[1]
        self.hpool = _Pool(self.H, self.dev)
        self.cpool = _Pool(self.H, self.dev)
        self.apool = _Pool(self.T, self.dev)
        self.w = decoder_w_struct(decoder_params(decoder))
        self._keep = None

    def seed(self, h, c):
        """Encoder outputs become pool rows 0..B-1."""
        B = h.shape[0]
        _, hv = self.hpool.reserve(B)
        _, cv = self.cpool.reserve(B)
        self.apool.reserve(B)
        hv.copy_(h)
        cv.copy_(c)

    def step(self, flat_obs, u_desc, state_rows, beam_indices, k):
        """Dictionary-style front end of `step_arrays`: flat_obs = N index-form observations; u_desc = N
        descriptors (row, view, heading, elevation) of the previous action's embedding or None (u_begin = zeros,
        model.py:368); state_rows = pool rows of (h, c); beam_indices = instance of each state.  Returns (pool
        row base of the N new states, a_num [N], top-k columns [N,k] and their log-probabilities [N,k])."""
        N = len(flat_obs)
        a_num = np.array([len(ob['adj_loc_list']) for ob in flat_obs], np.int64)
        A = int(a_num.max())
        head, elev = np.zeros((N, A)), np.zeros((N, A))
        cview = np.zeros((N, A), np.int64)
        for i, ob in enumerate(flat_obs):
            for a_, d in enumerate(ob['adj_loc_list'][1:], 1):
                cview[i, a_], head[i, a_], elev[i, a_] = d['absViewIndex'], d['rel_heading'], d['rel_elevation']
        has_u = np.array([u is not None for u in u_desc])
        ud = [u if u is not None else (0, 0, 0.0, 0.0) for u in u_desc]
        inputs = dict(vp=np.array([ob['vp_row'] for ob in flat_obs]), view=np.array([ob['viewIndex'] for ob in flat_obs]),
                      a_num=a_num, cand_view=cview, sincos=cand_sincos(head, elev), hrow=np.asarray(state_rows),
                      crow=np.asarray(beam_indices), has_u=has_u, u_vp=np.array([u[0] for u in ud]),
                      u_view=np.array([u[1] for u in ud]),
                      u_sincos=cand_sincos(np.array([u[2] for u in ud]), np.array([u[3] for u in ud])))
        base, idx, logp = self.step_arrays(inputs, k)
        return base, a_num, idx, logp

    def step_arrays(self, inp, k):
        """One decoder step over N states given as index arrays (see frontier._step_inputs): vp / view [N]
        (feature-table row, view index), a_num [N], cand_view [N,>=A], sincos [N,>=A,4], hrow / crow [N] (pool row
        of h and c, instruction row), has_u [N] and the (u_vp, u_view, u_sincos [N,4]) descriptor of the previous
        action's embedding.  Returns (pool row base of the N new states, top-k action columns [N,k] (-1 = none)
        and their log-probabilities [N,k], numpy); k = 0: the whole row."""
        st, dev = self.store, self.dev
        N = len(inp['vp'])
        A = int(inp['a_num'].max())
        k = min(k, A) if k else A
        # ONE upload: 8 int32 columns of N (each contiguous: no per-column copies on the device), the [N,2] and
        # [N,A] candidate views, then the float32 sin/cos blocks bit-cast into the same int32 buffer
        hu = inp['has_u']
        u_cv_h = np.zeros((N, 2), np.int32)
        u_cv_h[:, 1] = np.where(hu, inp['u_view'], 0)
        u_sc_h = np.zeros((N, 2, 4), np.float32)
        u_sc_h[:, 1] = np.where(hu[:, None], inp['u_sincos'], 0)
        parts = [np.asarray(inp['vp'], np.int32), np.asarray(inp['view'], np.int32), np.asarray(inp['a_num'], np.int32),
                 np.asarray(inp['hrow'], np.int32), np.asarray(inp['crow'], np.int32),
                 np.where(hu, inp['u_vp'], 0).astype(np.int32), hu.astype(np.int32), np.full(N, 2, np.int32),
                 u_cv_h.reshape(-1), np.ascontiguousarray(inp['cand_view'][:, :A], np.int32).reshape(-1),
                 u_sc_h.reshape(-1).view(np.int32),
                 np.ascontiguousarray(inp['sincos'][:, :A], np.float32).reshape(-1).view(np.int32)]
        di = torch.from_numpy(np.concatenate(parts)).to(dev)
        vp, view, a_num, hrow, crow, u_vp, u_act, two = (di[j * N:(j + 1) * N] for j in range(8))
        o = 8 * N
        u_cv = di[o:o + 2 * N].view(N, 2)
        cv = di[o + 2 * N:o + (2 + A) * N].view(N, A)
        o += (2 + A) * N
        u_sc = di[o:o + 8 * N].view(torch.float32).view(N, 2, 4)
        sc = di[o + 8 * N:o + (8 + 4 * A) * N].view(torch.float32).view(N, A, 4)
        df = None
        s = stream()
        new = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)  # noqa: E731
        h0, c0 = new(N, self.H), new(N, self.H)
        H = self.H
        gat = (_lib.RowMove * 2)(_lib.RowMove(self.hpool.buf.data_ptr(), h0.data_ptr(), hrow.data_ptr(), H, H, H, 0),
                                 _lib.RowMove(self.cpool.buf.data_ptr(), c0.data_ptr(), hrow.data_ptr(), H, H, H, 0))
        call('sf_move_rows', gat, 2, N, s)                 # h_t[flat_indices], c_t[flat_indices]: one launch
        ucand = st.cands(u_vp, u_cv, u_sc, two, 2)

        tape = decoder_tape(N, self.H, st.F, self.D, st.V, self.T, A, dev)
        # the previous action's embedding straight into the first half of the LSTM input rows (no copy in the step)
        call('sf_gather_actions_ld', byref(ucand), N, ptr(u_act), ptr(tape['xin']), 2 * st.F, s)
        base, tape['h1'] = self.hpool.reserve(N)
        _, tape['c1'] = self.cpool.reserve(N)
        _, tape['alpha'] = self.apool.reserve(N)
        pano = st.pano(vp, view)
        cnd = st.cands(vp, cv, sc, a_num, A)
        tp = tape_struct(tape)
        call('sf_attn_decoder_fwd', byref(self.w), byref(pano), byref(cnd), N, self.H, self.D, self.T,
             None, ptr(h0), ptr(c0), ptr(self.ctx), ptr(self.mask), ptr(crow), byref(tp), None,
             None, 0, *ws_args(dev))
        idx = torch.empty(N, k, dtype=torch.int32, device=dev)
        logp = new(N, k)
        call('sf_logprob_topk', ptr(tape['logit']), A, N, A, ptr(a_num), k, ptr(idx), ptr(logp), s)
        self._keep = (di, df, tape, h0, c0, gat, two, vp, view, a_num, hrow, crow, u_vp, u_act,
                      u_cv, cv, u_sc, sc)
        both = torch.cat((idx.to(torch.float32), logp), dim=1).cpu().numpy()       # ONE D2H copy per iteration
        return base, both[:, :k].astype(np.int64), both[:, k:]

    def step_logprobs(self, inp):
        """`step_arrays` over the whole candidate row: (pool row base, log-probabilities [N, A], -inf where the
        state has no such candidate)."""
        base, order, scores = self.step_arrays(inp, 0)
        logp = np.full(order.shape, -np.inf, np.float32)
        valid = order >= 0
        rows = np.broadcast_to(np.arange(len(order))[:, None], order.shape)
        logp[rows[valid], order[valid]] = scores[valid]
        return base, logp

    def attention_rows(self, rows):
        """Text-attention rows (host) for `attentions` in the result dictionaries."""
        if not rows:
            return []
        r = torch.tensor(rows, dtype=torch.int64, device=self.dev)
        return list(self.apool.buf[r].cpu().numpy())


class GraphStep:
    """The decoder step of the state-factored search as ONE hipGraph over fixed buffers (follower.py:783-836 per
    iteration: observations of the expanded states, `h_t[flat_indices]`, the AttnDecoderLSTM step, log_softmax).

    An iteration expands at most `successor_size` states per instance, so the step has a fixed capacity; what changes
    between iterations is a [8, cap] block of int32 (frontier_core.cpp: fill_inputs) written into pinned memory.  The
    graph: that block H2D; sf_nav_step twice (the states' and their parents' candidate lists straight from the device
    navigation table -- nothing but the eight integers per state is packed on the host); the previous action's
    embedding (sf_gather_actions over the parent's candidates; action 0 = zeros = u_begin); h / c rows from the state
    pool; sf_attn_decoder_fwd with per-state instruction rows; log_softmax in column order; h / c / attention rows of
    the new states scattered to their pool rows; the [cap, A] log-probabilities D2H into pinned memory.  One replay
    and one stream sync per iteration instead of ~25 host-issued launches, an upload and a blocking download.

    Instructions live in a [instances, t_max] buffer (padding masked out), so the graph outlives the minibatch: it is
    captured once per (decoder, store, navigation table, sizes) and re-captured only when a weight (or one of its
    cached layouts) moved or the state pool had to grow."""

    def __init__(self, decoder, store, nav, n_inst, cap, t_max, pool_rows=1 << 14):
        dev = store.device
        self.dec, self.store, self.nav, self.dev = decoder, store, nav, dev
        self.n_inst, self.cap, self.T, self.A = n_inst, cap, t_max, nav.A
        self.H = decoder.hidden_size
        self.D = decoder.visual_attention_layer.linear_in_h.weight.shape[0]
        f32 = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)           # noqa: E731
        i32 = lambda *s: torch.zeros(*s, device=dev, dtype=torch.int32)             # noqa: E731
        self.pin_in = torch.zeros(8, cap, dtype=torch.int32).pin_memory()
        self.inputs = self.pin_in.numpy()                                            # what fill_inputs writes
        self.dev_in = i32(8, cap)
        A = self.A
        # observations of the states [0] and of their parents [1]: halves of one [2 cap] look-up
        self.obs2 = dict(row=i32(2 * cap), vp=i32(2 * cap), view=i32(2 * cap), a_num=i32(2 * cap),
                         cand_view=i32(2 * cap, A), sincos=f32(2 * cap, A, 4))
        self.obs = [{k: v[j * cap:(j + 1) * cap] for k, v in self.obs2.items()} for j in range(2)]
        self.h0, self.c0 = f32(cap, self.H), f32(cap, self.H)
        self.tape = decoder_tape(cap, self.H, store.F, self.D, store.V, t_max, A, dev)
        self.logp = f32(cap, A)
        self.pin_out = torch.zeros(cap, A, dtype=torch.float32).pin_memory()
        self.out = self.pin_out.numpy()
        self.ctx = f32(n_inst, t_max, self.H)
        self.mask = torch.ones(n_inst, t_max, dtype=torch.uint8, device=dev)
        self.pool_rows = pool_rows
        self.hpool, self.cpool, self.apool = f32(pool_rows, self.H), f32(pool_rows, self.H), f32(pool_rows, t_max)
        self.n = 0
        self.graph = self.baked = self._stream = None

    # ---- per search
    def load(self, ctx, mask, h, c):
        """The minibatch's encoder outputs: instruction rows 0..B-1, pool rows 0..B-1."""
        B, T = ctx.shape[0], ctx.shape[1]
        if B > self.n_inst or T > self.T:
            raise ValueError('GraphStep built for %d instructions of <= %d tokens' % (self.n_inst, self.T))
        self.mask.fill_(1)
        self.mask[:B, :T].copy_(mask.to(torch.uint8))
        self.ctx[:B, :T].copy_(ctx)
        self.hpool[:B].copy_(h)
        self.cpool[:B].copy_(c)
        self.n = B
        w = bytes(decoder_w_struct(decoder_params(self.dec)))        # (also refreshes stale cached layouts in place)
        if self.graph is None or w != self.baked:
            self._capture()

    def _issue(self):
        st, nav, cap, A = self.store, self.nav, self.cap, self.A
        s = stream()
        self.dev_in.copy_(self.pin_in, non_blocking=True)
        act, hrow, crow, dst = (self.dev_in[j] for j in range(4, 8))
        ns = nav.struct()
        # rows 0-1 of the block: nav rows of the states, then of their parents; rows 2-3 their views (fill_inputs)
        o = self.obs2
        call('sf_nav_step', byref(ns), 2 * cap, ptr(self.dev_in[0]), ptr(self.dev_in[2]), None, None, None, 0, None,
             ptr(o['row']), ptr(o['vp']), ptr(o['view']), ptr(o['a_num']), ptr(o['cand_view']), ptr(o['sincos']), None, s)
        cur, par = self.obs
        ucand = st.cands(par['vp'], par['cand_view'], par['sincos'], par['a_num'], A)
        # (the previous action's embedding straight into the first half of the LSTM input rows: no copy in the step)
        call('sf_gather_actions_ld', byref(ucand), cap, ptr(act), ptr(self.tape['xin']), 2 * st.F, s)
        H = self.H
        gat = (_lib.RowMove * 2)(_lib.RowMove(self.hpool.data_ptr(), self.h0.data_ptr(), hrow.data_ptr(), H, H, H, 0),
                                 _lib.RowMove(self.cpool.data_ptr(), self.c0.data_ptr(), hrow.data_ptr(), H, H, H, 0))
        call('sf_move_rows', gat, 2, cap, s)
        pano = st.pano(cur['vp'], cur['view'])
        cnd = st.cands(cur['vp'], cur['cand_view'], cur['sincos'], cur['a_num'], A)
        w = decoder_w_struct(decoder_params(self.dec))
        tp = tape_struct(self.tape)
        call('sf_attn_decoder_fwd', byref(w), byref(pano), byref(cnd), cap, self.H, self.D, self.T,
             None, ptr(self.h0), ptr(self.c0), ptr(self.ctx), ptr(self.mask), ptr(crow), byref(tp), None,
             None, 0, *ws_args(self.dev))
        call('sf_logprob_topk', ptr(self.tape['logit']), A, cap, A, ptr(cur['a_num']), A, None, ptr(self.logp), s)
        sca = (_lib.RowMove * 3)(*(_lib.RowMove(src.data_ptr(), pool.data_ptr(), dst.data_ptr(), width, width, width, 1)
                                   for src, pool, width in ((self.tape['h1'], self.hpool, H), (self.tape['c1'], self.cpool, H),
                                                            (self.tape['alpha'], self.apool, self.T))))
        call('sf_move_rows', sca, 3, cap, s)
        self.pin_out.copy_(self.logp, non_blocking=True)
        self._keep = (ns, ucand, pano, cnd, w, tp, gat, sca)

    def _capture(self):
        self.pin_in.zero_()
        self.pin_in[7].fill_(-1)                                     # (a warm-up that writes no pool row)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            self._issue()                                            # warm-up: workspace, cached layouts
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with graph_capture(graph, side):
                self._issue()
        torch.cuda.current_stream().wait_stream(side)
        self.graph, self._stream = graph, side                       # (the graph bakes the capture stream's workspace)
        self.baked = bytes(decoder_w_struct(decoder_params(self.dec)))

    # ---- the native loop (sim/frontier_core.cpp: run_graph) launches the graph itself
    def native_loop_ready(self):
        return self.graph is not None and hasattr(self.graph, 'raw_cuda_graph_exec') and _hip_entry_points() is not None

    def native_launch_args(self):
        """(hipGraphLaunch, hipStreamSynchronize, graph exec handle, stream handle) as integers."""
        launch, sync = _hip_entry_points()
        return launch, sync, int(self.graph.raw_cuda_graph_exec()), int(torch.cuda.current_stream(self.dev).cuda_stream)

    # ---- per iteration
    def run(self, n):
        """Inputs are in `self.inputs` (columns 0..n-1; the rest padding with destination -1).  Returns the
        log-probabilities [cap, A] (host view, valid until the next run); the new states are pool rows
        self.n - n .. self.n - 1."""
        if self.n + n > self.pool_rows:
            self._grow(self.n + n)
        self.n += n
        self.graph.replay()
        torch.cuda.current_stream().synchronize()
        return self.out

    def _grow(self, need):
        rows = max(2 * self.pool_rows, need)
        for name in ('hpool', 'cpool', 'apool'):
            old = getattr(self, name)
            new = torch.zeros(rows, old.shape[1], device=self.dev, dtype=torch.float32)
            new[:self.n].copy_(old[:self.n])
            setattr(self, name, new)
        self.pool_rows = rows
        keep = self.pin_in.clone()
        self._capture()                                              # the graph holds the pools' addresses
        self.pin_in.copy_(keep)

    def attention_rows(self, rows):
        if not rows:
            return []
        r = torch.tensor(rows, dtype=torch.int64, device=self.dev)
        return list(self.apool[r].cpu().numpy())


_HIP_ENTRY = []


def _hip_entry_points():
    """Addresses of hipGraphLaunch / hipStreamSynchronize in the HIP runtime THIS process has loaded (the one torch and
    libsf_hip.so run on), for the native search loop -- or None."""
    if not _HIP_ENTRY:
        found = None
        try:
            path = None
            with open('/proc/self/maps') as f:
                for line in f:
                    if 'libamdhip64' in line:
                        path = line.split()[-1]
                        break
            if path:
                hip = C.CDLL(path)
                found = (C.cast(hip.hipGraphLaunch, C.c_void_p).value, C.cast(hip.hipStreamSynchronize, C.c_void_p).value)
        except (OSError, AttributeError):
            found = None
        _HIP_ENTRY.append(found)
    return _HIP_ENTRY[0]


def graph_step_for(agent, nav, n_inst, cap):
    """The agent's GraphStep for these sizes (built on first use, kept on the agent)."""
    t_max = agent.max_instruction_length
    # (a captured step keeps the gate-product kernel it was captured with: runtime.strict_gate_product)
    key = (id(agent.decoder), id(agent.store), id(nav), n_inst, cap, t_max, int(_lib.lib.sf_gate_product_is_strict()))
    cache = agent.__dict__.setdefault('_graph_steps', {})
    gs = cache.get(key)
    if gs is None or gs.dec is not agent.decoder or gs.store is not agent.store or gs.nav is not nav:
        cache.clear()                                                # (one live configuration: the pools are ~100 MB)
        gs = cache[key] = GraphStep(agent.decoder, agent.store, nav, n_inst, cap, t_max)
    return gs


def _require_store(agent):
    if getattr(agent, 'store', None) is None:
        raise RuntimeError('search runs on index-form observations: give the agent a '
                           'features.FeatureStore (agent.store)')
    if not hasattr(agent.env, 'panorama'):
        raise RuntimeError('search needs an env.R2RIndexEnv (index-form observations over the feature table)')


def _encode_items(agent, items):
    """Encoder pass over the minibatch's instructions: (ctx, mask, h_0, c_0), detached."""
    seq, seq_mask, seq_lengths = batch_instructions_from_encoded(
        [it['instr_encoding'] for it in items], agent.max_instruction_length, reverse=agent.reverse_instruction,
        device=agent._device())
    with torch.no_grad():
        ctx, h_t, c_t = agent.encoder(seq, seq_lengths)
    return ctx.detach(), seq_mask, h_t.detach(), c_t.detach()


class FlatSpeakerDecoder:
    """One SpeakerDecoderLSTM step (model.py:487-519) over a flat list of word hypotheses."""

    def __init__(self, decoder, ctx, path_mask):
        from .model import _SPK_TAPE
        self.dec, self.keys = decoder, _SPK_TAPE
        self.ctx = ctx.contiguous()
        self.mask = path_mask.to(torch.uint8).contiguous()
        self.dev = ctx.device
        self.H, self.E = decoder.hidden_size, decoder.embedding.weight.shape[1]
        self.vocab = decoder.decoder2action.weight.shape[0]
        self.ldv = (self.vocab + 3) & ~3
        self.Tp = ctx.shape[1]
        self.hpool, self.cpool, self.apool = _Pool(self.H, self.dev), _Pool(self.H, self.dev), _Pool(self.Tp, self.dev)
        self.w = decoder._w_struct()
        self._keep = None

    def seed(self, h, c):
        B = h.shape[0]
        _, hv = self.hpool.reserve(B)
        _, cv = self.cpool.reserve(B)
        self.apool.reserve(B)
        hv.copy_(h)
        cv.copy_(c)

    def step(self, words, rows, inst, k):
        """words / rows / inst [N]: previous word, pool row of (h, c), path of every hypothesis.  Returns (pool
        row base of the N new states, top-k words [N,k], their log-probabilities [N,k])."""
        dev, H = self.dev, self.H
        N = len(words)
        k = min(k, self.vocab)
        ints = torch.from_numpy(np.stack((rows, inst)).astype(np.int32)).to(dev)
        wd = torch.from_numpy(np.asarray(words, np.int64)).to(dev)
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        h0, c0 = new(N, H), new(N, H)
        s_ = stream()
        call('sf_gather_rows', ptr(self.hpool.buf), H, ptr(ints[0]), N, H, ptr(h0), H, s_)
        call('sf_gather_rows', ptr(self.cpool.buf), H, ptr(ints[0]), N, H, ptr(c0), H, s_)
        tape = dict(emb=new(N, self.E), gates=new(N, 4 * H), cat2=new(N, 2 * H), t_text=new(N, H),
                    h_tilde=new(N, H), logit=new(N, self.ldv))
        base, tape['h1'] = self.hpool.reserve(N)
        _, tape['c1'] = self.cpool.reserve(N)
        _, tape['alpha'] = self.apool.reserve(N)
        tp = _lib.SpkDecoderTape(*(tape[key].data_ptr() for key in self.keys))
        call('sf_speaker_decoder_fwd', byref(self.w), N, self.E, H, self.Tp, self.vocab, ptr(wd), ptr(h0), ptr(c0),
             ptr(self.ctx), ptr(self.mask), ptr(ints[1]), byref(tp), None, 0, *ws_args(dev))
        idx = torch.empty(N, k, dtype=torch.int32, device=dev)
        logp = new(N, k)
        call('sf_logprob_topk', ptr(tape['logit']), self.ldv, N, self.vocab, None, k, ptr(idx), ptr(logp), s_)
        self._keep = (ints, wd, h0, c0, tape)
        both = torch.cat((idx.to(torch.float32), logp), dim=1).cpu().numpy()
        return base, both[:, :k].astype(np.int64), both[:, k:]

    def attention_rows(self, rows):
        if not rows:
            return []
        return list(self.apool.buf[torch.tensor(rows, dtype=torch.int64, device=self.dev)].cpu().numpy())


def beam_search(agent, beam_size, load_next_minibatch=True, mask_undo=False):
    """follower.py:541-718.  Returns (trajs, completed, traversed_lists=None)."""
    from . import frontier
    return frontier.beam_search(agent, beam_size, load_next_minibatch, mask_undo)


def state_factored_search(agent, completion_size, successor_size, load_next_minibatch=True,
                          mask_undo=False, first_n_ws_key=4):
    """follower.py:720-980.  Returns (trajs, completed_list, traversed_lists)."""
    from . import frontier
    return frontier.state_factored_search(agent, completion_size, successor_size, load_next_minibatch, mask_undo,
                                          first_n_ws_key)


def speaker_beam_search(speaker, beam_size, path_obs, path_actions):
    """speaker.py:211-318."""
    from . import frontier
    return frontier.speaker_beam_search(speaker, beam_size, path_obs, path_actions)


# ------------------------------------------------------------------------------ pragmatic re-ranking
@gc_paused
def rational_mix(candidate_lists_by_instr_id, speaker_weight):
    """rational_follower.py:117-148: standardise follower and speaker scores over ALL candidates,
    pick per instruction the candidate maximising the weighted sum.  Returns (results, index counts)."""
    lists = list(candidate_lists_by_instr_id.values())
    follower_scores = np.array([c['follower_score'] for l in lists for c in l], np.float64)
    speaker_scores = np.array([c['speaker_score'] for l in lists for c in l], np.float64)
    speaker_std, follower_std = np.std(speaker_scores), np.std(follower_scores)
    sw = speaker_weight / speaker_std
    fw = (1 - speaker_weight) / follower_std
    mixed = speaker_scores * sw + follower_scores * fw           # (the reference's float64 expression, element-wise)
    results, index_count = {}, Counter()
    lo = 0
    for instr_id, candidates in candidate_lists_by_instr_id.items():
        best_ix = int(np.argmax(mixed[lo:lo + len(candidates)]))   # first maximum, like max() over enumerate
        lo += len(candidates)
        results[instr_id] = candidates[best_ix]
        index_count[best_ix] += 1
    return results, index_count


# ------------------------------------------------------------------------------ rational speaker
def generate_and_score_candidates(envir, speaker, follower, n_candidates, include_gold=False):
    """rational_speaker.py:9-106: for every gold path of the environment the speaker's `n_candidates` beam
    instructions (speaker.py:211-318), each scored by the follower with teacher forcing along that path
    (follower.py:342-428: how likely is THIS route given that instruction).  Returns {instr_id: [candidate, ...]},
    every candidate with `speaker_score`, `follower_score` and the follower's `actions`; stops when an instruction
    comes round again (one epoch)."""
    from .follower import EOS
    follower.env = envir
    speaker.env = envir
    envir.reset_epoch()
    speaker.feedback = 'argmax'
    for module in (follower.encoder, follower.decoder, speaker.encoder, speaker.decoder):
        module.eval()
    follower.set_beam_size(1)
    by_id = {}
    n_per_instance = []
    looped = False
    while not looped:
        path_obs, path_actions, gold_instr = envir.gold_obs_actions_and_instructions(speaker.max_episode_len,
                                                                                     load_next_minibatch=True)
        with torch.no_grad():
            gold = []
            if include_gold:
                gold, _ = speaker._score_obs_actions_and_instructions(path_obs, path_actions, gold_instr, 'teacher')
            beams = speaker.beam_search(n_candidates, path_obs, path_actions)
            if include_gold:
                assert len(gold) == len(beams)
                for g, bc in zip(gold, beams):
                    assert g['instr_id'] == bc[0]['instr_id']
                    bc.insert(0, g)
            cand_obs, cand_actions, cand_words = [], [], []
            for i, beam in enumerate(beams):
                n_per_instance.append(len(beam))
                for cand in beam:
                    cand_obs.append(path_obs[i])
                    cand_actions.append(path_actions[i])
                    idx = list(cand['word_indices'])
                    cand_words.append(idx[:-1] if idx and idx[-1] == EOS else idx)      # rational_speaker.py:63-65
            scored, _ = follower._score_obs_actions_and_instructions(cand_obs, cand_actions, cand_words)
        assert len(scored) == sum(len(b) for b in beams)
        k = 0
        for beam in beams:
            for cand in beam:
                fs = scored[k]
                k += 1
                assert cand['instr_id'] == fs['instr_id']
                cand['speaker_score'], cand['follower_score'], cand['actions'] = cand['score'], fs['score'], fs['actions']
            instr_id = beam[0]['instr_id']
            assert all(c['instr_id'] == instr_id for c in beam)
            if instr_id in by_id:
                looped = True
            else:
                by_id[instr_id] = beam
    return by_id


def predict_from_candidates(candidate_lists_by_instr_id, speaker_weights):
    """rational_speaker.py:109-137: scores standardised over ALL candidates; per weight and instruction the
    candidate with the highest weighted sum (first maximum).  Returns {weight: {instr_id: candidate}}."""
    return {w: rational_mix(candidate_lists_by_instr_id, w)[0] for w in speaker_weights}


def run_rational_speaker(envir, speaker_evaluator, speaker, follower, n_candidates, include_gold=False):
    """rational_speaker.py:140-165 without the file output: candidates, re-ranking for the 21 speaker weights 0, 0.05 ..
    1, and -- with an evaluator (eval_speaker.py, out of scope here) -- its score summary per weight.  Returns
    (scores_by_weight or None, results_by_weight)."""
    by_id = generate_and_score_candidates(envir, speaker, follower, n_candidates, include_gold)
    weights = [float(w) for w in np.arange(0, 20 + 1) / 20.0]
    results = predict_from_candidates(by_id, weights)
    scores = None
    if speaker_evaluator is not None:
        scores = {w: speaker_evaluator.score_results(r)[0] for w, r in results.items()}
    return scores, results


def _follower_candidates(follower, beam_size, include_gold, mask_undo, state_factored, key_fields):
    """One minibatch of candidate routes per instruction (rational_follower.py:35-62): optionally the gold route
    (a teacher-forced rollout) first, then the (state-factored) beam search's completions."""
    gold = []
    if include_gold:
        follower.feedback = 'teacher'
        gold = follower._rollout_with_loss()
    follower.feedback = 'argmax'
    if state_factored:
        cands, hyps, walks = follower.state_factored_search(beam_size, 1, load_next_minibatch=not include_gold,
                                                            mask_undo=mask_undo, first_n_ws_key=key_fields)
    else:
        cands, hyps, walks = follower.beam_search(beam_size, load_next_minibatch=not include_gold, mask_undo=mask_undo)
    if include_gold:
        assert len(gold) == len(cands)
        for g, c in zip(gold, cands):
            assert g['instr_id'] == c[0]['instr_id']
            c.insert(0, g)
    return cands, hyps, walks


def _walked_route(walk_so_far, hyp):
    """What the agent physically walks to END at `hyp` after the search (rational_follower.py:87-96): the
    search's own traversal, then from its last state to the candidate's end state."""
    tail = least_common_viewpoint_path(walk_so_far[-1], hyp)[1:]
    return [path_element_from_observation(s.observation) for s in list(walk_so_far) + tail]


def run_rational_follower(envir, evaluator, follower, speaker, beam_size, include_gold=False,
                          compute_oracle=False, mask_undo=False, state_factored_search=False,
                          state_first_n_ws_key=4, physical_traversal=False,
                          speaker_weights=(0., 0.95)):
    """rational_follower.py:12-190 without the file outputs: follower candidates by (state-factored)
    beam search, every candidate route scored by the speaker with teacher forcing (how likely is THIS
    instruction given that route), candidates re-ranked by `rational_mix` for every speaker weight.
    `evaluator` (eval.py, out of scope here) is optional: with one, returns its score summaries per
    weight like the reference; without, the chosen candidates per weight."""
    follower.env = envir
    envir.reset_epoch()
    for module in (follower.encoder, follower.decoder, speaker.encoder, speaker.decoder):
        module.eval()
    follower.set_beam_size(beam_size)
    # the search hands its routes to the speaker in index form the moment its last iteration is done: the device scores
    # them while the host builds their result dictionaries (a gold route in front of every list changes the batch: off)
    follower.candidates_hook = (speaker.route_scores_hook('teacher')
                                if state_factored_search and not include_gold and hasattr(speaker, 'route_scores_hook')
                                else None)
    by_instruction = {}
    while True:                                         # one epoch: until an instruction comes round again
        cands, hyps, walks = _follower_candidates(follower, beam_size, include_gold, mask_undo, state_factored_search,
                                                  state_first_n_ws_key)
        flat = flatten(cands)
        with torch.no_grad():
            spoken, _ = speaker._score_obs_actions_and_instructions(
                [c['observations'] for c in flat], [c['actions'] for c in flat], [c['instr_encoding'] for c in flat],
                feedback='teacher')
        assert len(spoken) == len(flat)
        for c, sp in zip(flat, spoken):
            assert c['instr_id'] == sp['instr_id']
            c['follower_score'], c['speaker_score'] = c['score'], sp['score']
            del c['observations']
        wrapped = False
        for b, group in enumerate(cands):
            if physical_traversal:
                offset = 1 if include_gold else 0       # the gold route is not a search hypothesis
                for c, hyp in zip(group[offset:], hyps[b]):
                    route = _walked_route(walks[b], hyp)
                    assert route[-1][0] == c['trajectory'][-1][0]
                    c['trajectory'] = route
            if compute_oracle and evaluator is not None:
                for c in group:
                    c['eval_result'] = evaluator._score_item(c['instr_id'], c['trajectory'])._asdict()
            instr_id = group[0]['instr_id']
            assert all(c['instr_id'] == instr_id for c in group)
            if instr_id in by_instruction:
                wrapped = True
            else:
                by_instruction[instr_id] = group
        if wrapped:
            break
    follower.candidates_hook = None
    outcome, picks = {}, {}
    for w in speaker_weights:
        chosen, picks[w] = rational_mix(by_instruction, w)
        outcome[w] = evaluator.score_results(chosen)[0] if evaluator is not None else chosen
    return outcome, picks
