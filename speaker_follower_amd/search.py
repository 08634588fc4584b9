"""Search procedures of the follower and the speaker on the HIP path:

* `beam_search`            -- Seq2SeqAgent.beam_search            (follower.py:541-718)
* `state_factored_search`  -- Seq2SeqAgent.state_factored_search  (follower.py:720-980)
* `speaker_beam_search`    -- Seq2SeqSpeaker.beam_search          (speaker.py:211-318)
* `rational_mix`, `run_rational_follower` -- the pragmatic re-ranking (rational_follower.py:12-190)

The bookkeeping (beams, caches, back-pointers, the environment walk) is host logic exactly where the
reference has it.  Every iteration is ONE flat decoder step over all live states on the device:

  reference (per iteration)                            here
  -----------------------------------------------     ------------------------------------------------
  np.stack of N x 36 x 2176 features + H2D             N x (row, view) int32 -> kernels gather from HBM
  all_u_t [N,A,2176] built on the host + H2D           N x A (view, sin/cos) -> gathered in the kernel
  h_t[flat_indices], c_t[flat_indices]                 sf_gather_rows from the state pool
  ctx[beam_indices], seq_mask[beam_indices] (copies)   ctx_row indirection inside the attention kernel
  log_softmax + topk + gather, .data on the host       sf_logprob_topk, one D2H of [N,k] per iteration
  last_action_embedding = all_u_t[i, a] (2176 floats)  (row, view, heading, elevation) descriptor

Observations must be in index form (`vp_row`, `viewIndex`, `adj_loc_list`; env.R2RIndexEnv) and the
agent needs a `features.FeatureStore` (`agent.store`).
"""
import ctypes as C
import heapq
import itertools
from collections import namedtuple, Counter

import numpy as np
import torch

from . import _lib
from ._lib import call
from .features import cand_sincos
from .follower import batch_instructions_from_encoded, EOS, BOS
from .model import decoder_params, decoder_w_struct, decoder_tape, tape_struct
from .runtime import ptr, stream, ws_args

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
        """flat_obs: N index-form observations; u_desc: N descriptors (row, view, heading, elevation)
        of the previous action's embedding or None (u_begin = zeros, model.py:368); state_rows: pool
        rows of (h, c); beam_indices: instance of each state.  Returns (pool row base of the N new
        states, a_num [N], top-k columns [N,k] and their log-probabilities [N,k] as numpy)."""
        st, dev = self.store, self.dev
        N = len(flat_obs)
        A = max(len(ob['adj_loc_list']) for ob in flat_obs)
        k = min(k, A) if k else A
        ints = np.zeros((N, 7 + 2 + A), np.int32)      # vp view a_num hrow crow u_vp u_act | u_cv[2] | cv[A]
        flts = np.zeros((N, 8 + 4 * A), np.float32)    # u_sincos[2,4] | sincos[A,4]
        head = np.zeros((N, A), np.float64)
        elev = np.zeros((N, A), np.float64)
        uh = np.zeros(N, np.float64)
        ue = np.zeros(N, np.float64)
        for i, ob in enumerate(flat_obs):
            adj = ob['adj_loc_list']
            r = ints[i]
            r[0], r[1], r[2], r[3], r[4] = ob['vp_row'], ob['viewIndex'], len(adj), state_rows[i], \
                beam_indices[i]
            for a in range(1, len(adj)):
                d = adj[a]
                r[9 + a] = d['absViewIndex']
                head[i, a], elev[i, a] = d['rel_heading'], d['rel_elevation']
            u = u_desc[i]
            if u is not None:
                r[5], r[6], r[8] = u[0], 1, u[1]
                uh[i], ue[i] = u[2], u[3]
        flts[:, 8:] = cand_sincos(head, elev).reshape(N, 4 * A)
        flts[:, 4:8] = cand_sincos(uh, ue)
        di = torch.from_numpy(ints).to(dev)
        df = torch.from_numpy(flts).to(dev)
        col = lambda j: di[:, j].contiguous()  # noqa: E731
        vp, view, a_num, hrow, crow, u_vp, u_act = (col(j) for j in range(7))
        u_cv = di[:, 7:9].contiguous()
        cv = di[:, 9:].contiguous()
        u_sc = df[:, :8].contiguous().view(N, 2, 4)
        sc = df[:, 8:].contiguous().view(N, A, 4)
        two = torch.full((N,), 2, dtype=torch.int32, device=dev)

        s = stream()
        new = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)  # noqa: E731
        h0, c0, u_prev = new(N, self.H), new(N, self.H), new(N, st.F)
        call('sf_gather_rows', ptr(self.hpool.buf), self.H, ptr(hrow), N, self.H, ptr(h0), self.H, s)
        call('sf_gather_rows', ptr(self.cpool.buf), self.H, ptr(hrow), N, self.H, ptr(c0), self.H, s)
        ucand = st.cands(u_vp, u_cv, u_sc, two, 2)
        call('sf_gather_actions', byref(ucand), N, ptr(u_act), ptr(u_prev), s)

        tape = decoder_tape(N, self.H, st.F, self.D, st.V, self.T, A, dev)
        base, tape['h1'] = self.hpool.reserve(N)
        _, tape['c1'] = self.cpool.reserve(N)
        _, tape['alpha'] = self.apool.reserve(N)
        pano = st.pano(vp, view)
        cnd = st.cands(vp, cv, sc, a_num, A)
        tp = tape_struct(tape)
        call('sf_attn_decoder_fwd', byref(self.w), byref(pano), byref(cnd), N, self.H, self.D, self.T,
             ptr(u_prev), ptr(h0), ptr(c0), ptr(self.ctx), ptr(self.mask), ptr(crow), byref(tp), None,
             None, 0, *ws_args(dev))
        idx = torch.empty(N, k, dtype=torch.int32, device=dev)
        logp = new(N, k)
        call('sf_logprob_topk', ptr(tape['logit']), A, N, A, ptr(a_num), k, ptr(idx), ptr(logp), s)
        self._keep = (di, df, tape, h0, c0, u_prev, two, vp, view, a_num, hrow, crow, u_vp, u_act,
                      u_cv, cv, u_sc, sc)
        return base, ints[:, 2].copy(), idx.cpu().numpy(), logp.cpu().numpy()

    def attention_rows(self, rows):
        """Text-attention rows (host) for `attentions` in the result dictionaries."""
        if not rows:
            return []
        r = torch.tensor(rows, dtype=torch.int64, device=self.dev)
        return list(self.apool.buf[r].cpu().numpy())


def _u_descriptor(ob, action):
    d = ob['adj_loc_list'][action]
    return (ob['vp_row'], d['absViewIndex'], d['rel_heading'], d['rel_elevation'])


def _require_store(agent):
    if getattr(agent, 'store', None) is None:
        raise RuntimeError('search runs on index-form observations: give the agent a '
                           'features.FeatureStore (agent.store)')


def _require_index_form(agent, obs0):
    if 'vp_row' not in obs0:
        raise RuntimeError("search needs observations with 'vp_row' (env.R2RIndexEnv)")


def _encode(agent, obs):
    enc = [o[0]['instr_encoding'] for o in obs]
    seq, seq_mask, seq_lengths = batch_instructions_from_encoded(
        enc, agent.max_instruction_length, reverse=agent.reverse_instruction, device=agent._device())
    with torch.no_grad():
        ctx, h_t, c_t = agent.encoder(seq, seq_lengths)
    return ctx.detach(), seq_mask, h_t.detach(), c_t.detach()


def _trajs(fd, completed_lists):
    """follower.py:694-716 / 953-975: result dictionaries from the final inference states."""
    rows = sorted({r for lst in completed_lists for s in lst
                   for r in backchain_inference_states(s)[4] if r is not None})
    att = dict(zip(rows, fd.attention_rows(rows)))
    trajs = []
    for this_completed in completed_lists:
        assert this_completed
        this_trajs = []
        for inf_state in this_completed:
            _, path_obs, path_actions, path_scores, path_att = backchain_inference_states(inf_state)
            this_trajs.append({
                'instr_id': path_obs[0]['instr_id'],
                'instr_encoding': path_obs[0]['instr_encoding'],
                'trajectory': [path_element_from_observation(ob) for ob in path_obs],
                'observations': path_obs,
                'actions': path_actions,
                'score': inf_state.score,
                'scores': path_scores,
                'attentions': [att[r] for r in path_att],
            })
        trajs.append(this_trajs)
    return trajs


def beam_search(agent, beam_size, load_next_minibatch=True, mask_undo=False):
    """follower.py:541-718.  Returns (trajs, completed, traversed_lists=None)."""
    env = agent.env
    _require_store(agent)
    assert env.beam_size >= beam_size
    world_states = env.reset(sort=True, beamed=True, load_next_minibatch=load_next_minibatch)
    obs = env.observe(world_states, beamed=True)
    batch_size = len(world_states)
    _require_index_form(agent, obs[0][0])
    ctx, seq_mask, h_t, c_t = _encode(agent, obs)
    fd = FlatDecoder(agent.decoder, agent.store, ctx, seq_mask)
    fd.seed(h_t, c_t)

    completed = [[] for _ in range(batch_size)]
    beams = [[InferenceState(prev_inference_state=None, world_state=ws[0], observation=o[0],
                             flat_index=i, last_action=-1, last_action_embedding=None,
                             action_count=0, score=0.0, h_t=i, c_t=i, last_alpha=None)]
             for i, (ws, o) in enumerate(zip(world_states, obs))]

    for t in range(agent.episode_len):
        flat_states = flatten(beams)
        beam_indices = [bi for bi, beam in enumerate(beams) for _ in beam]
        base, a_num, action_indices, action_scores = fd.step(
            flatten(obs), [s.last_action_embedding for s in flat_states],
            [s.h_t for s in flat_states], beam_indices, beam_size)

        start_index = 0
        all_successors = []
        for beam_index, (beam, beam_world_states, beam_obs) in enumerate(zip(beams, world_states, obs)):
            successors = []
            assert len(beam_world_states) == len(beam) == len(beam_obs)
            for inf_index, (inf_state, world_state, ob) in enumerate(zip(beam, beam_world_states,
                                                                         beam_obs)):
                flat_index = start_index + inf_index
                for action_score, action_index in zip(action_scores[flat_index],
                                                      action_indices[flat_index]):
                    if action_index < 0 or action_index >= a_num[flat_index]:          # is_valid == 0
                        continue
                    action_index = int(action_index)
                    successors.append(InferenceState(
                        prev_inference_state=inf_state, world_state=world_state, observation=ob,
                        flat_index=flat_index, last_action=action_index,
                        last_action_embedding=_u_descriptor(ob, action_index),
                        action_count=inf_state.action_count + 1,
                        score=_add_f32(inf_state.score, action_score),
                        h_t=base + flat_index, c_t=base + flat_index, last_alpha=base + flat_index))
            start_index += len(beam)
            successors = sorted(successors, key=lambda s: s.score, reverse=True)[:beam_size]
            all_successors.append(successors)

        succ_ws = [[s.world_state for s in succ] for succ in all_successors]
        succ_actions = [[s.last_action for s in succ] for succ in all_successors]
        succ_last_obs = [[s.observation for s in succ] for succ in all_successors]
        succ_ws = env.step(succ_ws, succ_actions, succ_last_obs, beamed=True)
        succ_obs = env.observe(succ_ws, beamed=True)
        all_successors = [[s._replace(world_state=w, observation=o) for s, w, o in zip(sl, wl, ol)]
                          for sl, wl, ol in zip(all_successors, succ_ws, succ_obs)]

        new_beams = []
        for beam_index, successors in enumerate(all_successors):
            new_beam = []
            for successor in successors:
                if successor.last_action == 0 or t == agent.episode_len - 1:
                    completed[beam_index].append(successor)
                else:
                    new_beam.append(successor)
            if len(completed[beam_index]) >= beam_size:
                new_beam = []
            new_beams.append(new_beam)
        beams = new_beams
        world_states = [[s.world_state for s in beam] for beam in beams]
        obs = [[s.observation for s in beam] for beam in beams]
        if not any(beam for beam in beams):
            break

    completed_sorted = [sorted(c, key=lambda s: s.score, reverse=True)[:beam_size] for c in completed]
    return _trajs(fd, completed_sorted), completed, None


def state_factored_search(agent, completion_size, successor_size, load_next_minibatch=True,
                          mask_undo=False, first_n_ws_key=4):
    """follower.py:720-980.  Returns (trajs, completed_list, traversed_lists)."""
    env = agent.env
    _require_store(agent)
    assert env.beam_size >= successor_size
    world_states = env.reset(sort=True, beamed=True, load_next_minibatch=load_next_minibatch)
    initial_obs = env.observe(world_states, beamed=True)
    batch_size = len(world_states)
    _require_index_form(agent, initial_obs[0][0])
    ctx, seq_mask, h_t, c_t = _encode(agent, initial_obs)
    fd = FlatDecoder(agent.decoder, agent.store, ctx, seq_mask)
    fd.seed(h_t, c_t)

    completed = [{} for _ in range(batch_size)]
    completed_holding = [{} for _ in range(batch_size)]
    state_cache = [
        {ws[0][0:first_n_ws_key]: (InferenceState(
            prev_inference_state=None, world_state=ws[0], observation=o[0], flat_index=None,
            last_action=-1, last_action_embedding=None, action_count=0, score=0.0, h_t=i, c_t=i,
            last_alpha=None), True)}
        for i, (ws, o) in enumerate(zip(world_states, initial_obs))]
    beams = [[inf_state for _, (inf_state, expanded) in sorted(cache.items())]
             for cache in state_cache]

    last_expanded_list, traversed_lists = [], []
    for beam in beams:
        assert len(beam) == 1
        last_expanded_list.append(beam[0])
        traversed_lists.append([beam[0]])

    def update_traversed_lists(new_visited_inf_states):
        assert len(new_visited_inf_states) == len(last_expanded_list) == len(traversed_lists)
        for instance_index, instance_states in enumerate(new_visited_inf_states):
            last_expanded = last_expanded_list[instance_index]
            assert last_expanded.world_state.viewpointId == \
                traversed_lists[instance_index][-1].world_state.viewpointId
            for inf_state in instance_states:
                path = least_common_viewpoint_path(last_expanded, inf_state)
                assert path[0].world_state.viewpointId == last_expanded.world_state.viewpointId
                assert path[-1].world_state.viewpointId == inf_state.world_state.viewpointId
                traversed_lists[instance_index].extend(path[1:])
                last_expanded = inf_state
            last_expanded_list[instance_index] = last_expanded

    while any(len(comp) < completion_size for comp in completed):
        flat_states = flatten(beams)
        beam_indices = [bi for bi, beam in enumerate(beams) for _ in beam]
        flat_obs = [s.observation for s in flat_states]
        base, a_num, order, order_scores = fd.step(
            flat_obs, [s.last_action_embedding for s in flat_states], [s.h_t for s in flat_states],
            beam_indices, 0)                                  # whole row (follower.py:802 topk(A))
        A = order.shape[1]
        log_probs = np.full((len(flat_states), A), -np.inf, np.float32)
        rows = np.arange(len(flat_states))[:, None]
        valid = order >= 0
        log_probs[np.broadcast_to(rows, order.shape)[valid], order[valid]] = order_scores[valid]

        start_index = 0
        all_successors = []
        for beam_index, (beam, beam_world_states) in enumerate(zip(beams, world_states)):
            successors = []
            assert len(beam_world_states) == len(beam)
            for inf_index, (inf_state, world_state) in enumerate(zip(beam, beam_world_states)):
                flat_index = start_index + inf_index
                ob = flat_obs[flat_index]
                for action_index in range(int(a_num[flat_index])):
                    successors.append(InferenceState(
                        prev_inference_state=inf_state, world_state=world_state, observation=ob,
                        flat_index=None, last_action=action_index,
                        last_action_embedding=_u_descriptor(ob, action_index),
                        action_count=inf_state.action_count + 1,
                        score=_add_f32(inf_state.score, log_probs[flat_index, action_index]),
                        h_t=base + flat_index, c_t=base + flat_index, last_alpha=base + flat_index))
            start_index += len(beam)
            all_successors.append(sorted(successors, key=lambda s: s.score, reverse=True))

        succ_ws = [[s.world_state for s in succ] for succ in all_successors]
        succ_actions = [[s.last_action for s in succ] for succ in all_successors]
        succ_last_obs = [[s.observation for s in succ] for succ in all_successors]
        succ_ws = env.step(succ_ws, succ_actions, succ_last_obs, beamed=True)
        all_successors = [[s._replace(world_state=w) for s, w in zip(sl, wl)]
                          for sl, wl in zip(all_successors, succ_ws)]
        assert len(all_successors) == len(state_cache)

        new_beams = []
        for beam_index, (successors, instance_cache) in enumerate(zip(all_successors, state_cache)):
            instance_completed = completed[beam_index]
            instance_completed_holding = completed_holding[beam_index]
            if len(instance_completed) >= completion_size:
                new_beams.append([])
                continue
            for successor in successors:
                ws_keys = successor.world_state[0:first_n_ws_key]
                if successor.last_action == 0 or successor.action_count == agent.episode_len:
                    if ws_keys not in instance_completed_holding or \
                            instance_completed_holding[ws_keys][0].score < successor.score:
                        instance_completed_holding[ws_keys] = (successor, False)
                else:
                    if ws_keys not in instance_cache or \
                            instance_cache[ws_keys][0].score < successor.score:
                        instance_cache[ws_keys] = (successor, False)

            uncompleted = ((k_, s, False) for (k_, (s, expanded)) in instance_cache.items()
                           if not expanded)
            done = ((k_, s, True) for (k_, (s, expanded)) in instance_completed_holding.items()
                    if not expanded)
            best = heapq.nlargest(successor_size, itertools.chain(uncompleted, done),
                                  key=lambda pair: pair[1].score)
            new_beam = []
            for ws_keys, inf_state, is_completed in best:
                if is_completed:
                    assert instance_completed_holding[ws_keys] == (inf_state, False)
                    instance_completed_holding[ws_keys] = (inf_state, True)
                    if ws_keys not in instance_completed or \
                            instance_completed[ws_keys].score < inf_state.score:
                        instance_completed[ws_keys] = inf_state
                else:
                    instance_cache[ws_keys] = (inf_state, True)
                    new_beam.append(inf_state)
            new_beams.append([] if len(instance_completed) >= completion_size else new_beam)

        beams = new_beams
        if not any(beam for beam in beams):
            break
        world_states = [[s.world_state for s in beam] for beam in beams]
        succ_obs = env.observe(world_states, beamed=True)
        beams = [[s._replace(observation=o) for s, o in zip(beam, ol)]
                 for beam, ol in zip(beams, succ_obs)]
        update_traversed_lists(beams)

    completed_list = [sorted(c.values(), key=lambda s: s.score, reverse=True)[:completion_size]
                      for c in completed]
    completed_ws = [[s.world_state for s in comp] for comp in completed_list]
    completed_obs = env.observe(completed_ws, beamed=True)
    completed_list = [[s._replace(observation=o) for s, o in zip(comp, ol)]
                      for comp, ol in zip(completed_list, completed_obs)]
    update_traversed_lists(completed_list)
    return _trajs(fd, completed_list), completed_list, traversed_lists


# ---------------------------------------------------------------------------------------- speaker
def speaker_backchain(last_inference_state):
    """speaker.py:20-32."""
    word_indices, scores, attentions = [], [], []
    inf_state, last_score = last_inference_state, None
    while inf_state is not None:
        word_indices.append(inf_state.last_word)
        attentions.append(inf_state.last_alpha)
        if last_score is not None:
            scores.append(last_score - inf_state.score)
        last_score = inf_state.score
        inf_state = inf_state.prev_inference_state
    scores.append(last_score)
    return word_indices[::-1][1:], scores[::-1][1:], attentions[::-1][1:]


def speaker_beam_search(speaker, beam_size, path_obs, path_actions):
    """speaker.py:211-318.  One flat SpeakerDecoderLSTM step per word over all live hypotheses."""
    from .model import _SPK_TAPE
    assert len(path_obs) == len(path_actions)
    start_obs, feats, acts, path_mask, _, _, perm_indices = \
        speaker._batch_observations_and_actions(path_obs, path_actions, None)
    batch_size = len(start_obs)
    dec = speaker.decoder
    dev = speaker._device()
    with torch.no_grad():
        ctx, h_t, c_t = speaker.encoder(acts, feats)
    ctx = ctx.detach().contiguous()
    mask = path_mask.to(torch.uint8).contiguous()
    H, E = dec.hidden_size, dec.embedding.weight.shape[1]
    vocab = dec.decoder2action.weight.shape[0]
    ldv = (vocab + 3) & ~3
    Tp = ctx.shape[1]
    hpool, cpool, apool = _Pool(H, dev), _Pool(H, dev), _Pool(Tp, dev)
    _, hv = hpool.reserve(batch_size)
    _, cv = cpool.reserve(batch_size)
    apool.reserve(batch_size)
    hv.copy_(h_t.detach())
    cv.copy_(c_t.detach())
    w = dec._w_struct()
    k = min(beam_size, vocab)
    new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731

    completed = [[] for _ in range(batch_size)]
    beams = [[SpeakerInferenceState(None, i, BOS, 0, 0.0, None)] for i in range(batch_size)]
    for t in range(speaker.instruction_len):
        flat = flatten(beams)
        N = len(flat)
        beam_indices = [bi for bi, beam in enumerate(beams) for _ in beam]
        ints = torch.tensor([[s.flat_index for s in flat], beam_indices], dtype=torch.int32, device=dev)
        words = torch.tensor([s.last_word for s in flat], dtype=torch.int64, device=dev)
        h0, c0 = new(N, H), new(N, H)
        s_ = stream()
        call('sf_gather_rows', ptr(hpool.buf), H, ptr(ints[0]), N, H, ptr(h0), H, s_)
        call('sf_gather_rows', ptr(cpool.buf), H, ptr(ints[0]), N, H, ptr(c0), H, s_)
        tape = dict(emb=new(N, E), gates=new(N, 4 * H), cat2=new(N, 2 * H), t_text=new(N, H),
                    h_tilde=new(N, H), logit=new(N, ldv))
        base, tape['h1'] = hpool.reserve(N)
        _, tape['c1'] = cpool.reserve(N)
        _, tape['alpha'] = apool.reserve(N)
        tp = _lib.SpkDecoderTape(*(tape[key].data_ptr() for key in _SPK_TAPE))
        call('sf_speaker_decoder_fwd', byref(w), N, E, H, Tp, vocab, ptr(words), ptr(h0), ptr(c0),
             ptr(ctx), ptr(mask), ptr(ints[1]), byref(tp), None, 0, *ws_args(dev))
        idx = torch.empty(N, k, dtype=torch.int32, device=dev)
        logp = new(N, k)
        call('sf_logprob_topk', ptr(tape['logit']), ldv, N, vocab, None, k, ptr(idx), ptr(logp), s_)
        word_indices, word_scores = idx.cpu().numpy(), logp.cpu().numpy()

        start_index = 0
        all_successors = []
        for beam in beams:
            successors = []
            for inf_index, inf_state in enumerate(beam):
                flat_index = start_index + inf_index
                for word_score, word_index in zip(word_scores[flat_index], word_indices[flat_index]):
                    successors.append(SpeakerInferenceState(
                        inf_state, base + flat_index, int(word_index), inf_state.word_count + 1,
                        _add_f32(inf_state.score, word_score), base + flat_index))
            start_index += len(beam)
            all_successors.append(sorted(successors, key=lambda s: s.score, reverse=True)[:beam_size])

        new_beams = []
        for beam_index, successors in enumerate(all_successors):
            new_beam = []
            for successor in successors:
                if successor.last_word == EOS or t == speaker.instruction_len - 1:
                    completed[beam_index].append(successor)
                else:
                    new_beam.append(successor)
            if len(completed[beam_index]) >= beam_size:
                new_beam = []
            new_beams.append(new_beam)
        beams = new_beams
        if not any(beam for beam in beams):
            break

    tok = getattr(speaker.env, 'tokenizer', None)
    outputs = [[] for _ in range(batch_size)]
    for perm_index, src_index in enumerate(perm_indices):
        this_outputs = outputs[src_index]
        assert len(this_outputs) == 0
        instr_id = start_obs[perm_index]['instr_id']
        best = sorted(completed[perm_index], key=lambda s: s.score, reverse=True)[:beam_size]
        for inf_state in best:
            word_idx, scores, att_rows = speaker_backchain(inf_state)
            rows = torch.tensor(att_rows, dtype=torch.int64, device=dev)
            this_outputs.append({
                'instr_id': instr_id,
                'word_indices': word_idx,
                'score': inf_state.score,
                'scores': scores,
                'words': (tok.decode_sentence(word_idx, break_on_eos=True, join=False)
                          if tok is not None else list(word_idx)),
                'attentions': list(apool.buf[rows].cpu().numpy()),
            })
    return outputs


# ------------------------------------------------------------------------------ pragmatic re-ranking
def rational_mix(candidate_lists_by_instr_id, speaker_weight):
    """rational_follower.py:117-148: standardise follower and speaker scores over ALL candidates,
    pick per instruction the candidate maximising the weighted sum.  Returns (results, index counts)."""
    follower_scores = [c['follower_score'] for l in candidate_lists_by_instr_id.values() for c in l]
    speaker_scores = [c['speaker_score'] for l in candidate_lists_by_instr_id.values() for c in l]
    speaker_std, follower_std = np.std(speaker_scores), np.std(follower_scores)
    sw = speaker_weight / speaker_std
    fw = (1 - speaker_weight) / follower_std
    results, index_count = {}, Counter()
    for instr_id, candidates in candidate_lists_by_instr_id.items():
        best_ix, best_cand = max(enumerate(candidates),
                                 key=lambda tp: tp[1]['speaker_score'] * sw + tp[1]['follower_score'] * fw)
        results[instr_id] = best_cand
        index_count[best_ix] += 1
    return results, index_count


def run_rational_follower(envir, evaluator, follower, speaker, beam_size, include_gold=False,
                          compute_oracle=False, mask_undo=False, state_factored_search=False,
                          state_first_n_ws_key=4, physical_traversal=False,
                          speaker_weights=(0., 0.95)):
    """rational_follower.py:12-190 without the file outputs: follower candidates by (state-factored)
    beam search, scored by the speaker with teacher forcing, re-ranked by `rational_mix`.
    `evaluator` (eval.py, out of scope here) is optional: with one, returns its score summaries per
    weight like the reference; without, the chosen candidates per weight."""
    follower.env = envir
    envir.reset_epoch()
    for m in (follower.encoder, follower.decoder, speaker.encoder, speaker.decoder):
        m.eval()
    follower.set_beam_size(beam_size)
    candidate_lists_by_instr_id = {}
    looped = False
    while True:
        if include_gold:
            follower.feedback = 'teacher'
            gold_candidates = follower._rollout_with_loss()
        else:
            gold_candidates = []
        follower.feedback = 'argmax'
        if state_factored_search:
            beam_candidates, candidate_inf_states, traversed_lists = follower.state_factored_search(
                beam_size, 1, load_next_minibatch=not include_gold, mask_undo=mask_undo,
                first_n_ws_key=state_first_n_ws_key)
        else:
            beam_candidates, candidate_inf_states, traversed_lists = follower.beam_search(
                beam_size, load_next_minibatch=not include_gold, mask_undo=mask_undo)
        if include_gold:
            assert len(gold_candidates) == len(beam_candidates)
            for i, bc in enumerate(beam_candidates):
                assert gold_candidates[i]['instr_id'] == bc[0]['instr_id']
                bc.insert(0, gold_candidates[i])

        cands = flatten(beam_candidates)
        with torch.no_grad():
            scored, _ = speaker._score_obs_actions_and_instructions(
                [c['observations'] for c in cands], [c['actions'] for c in cands],
                [c['instr_encoding'] for c in cands], feedback='teacher')
        assert len(scored) == len(cands)
        start_index = 0
        for instance_index, instance_candidates in enumerate(beam_candidates):
            for i, candidate in enumerate(instance_candidates):
                sc = scored[start_index + i]
                assert candidate['instr_id'] == sc['instr_id']
                candidate['follower_score'] = candidate['score']
                candidate['speaker_score'] = sc['score']
                del candidate['observations']
                if physical_traversal:
                    last_traversed = traversed_lists[instance_index][-1]
                    cand_state = candidate_inf_states[instance_index][i]
                    path = least_common_viewpoint_path(last_traversed, cand_state)
                    inf_traj = traversed_lists[instance_index] + path[1:]
                    physical = [path_element_from_observation(s.observation) for s in inf_traj]
                    assert physical[-1][0] == candidate['trajectory'][-1][0]
                    candidate['trajectory'] = physical
                if compute_oracle and evaluator is not None:
                    candidate['eval_result'] = evaluator._score_item(
                        candidate['instr_id'], candidate['trajectory'])._asdict()
            start_index += len(instance_candidates)
            instr_id = instance_candidates[0]['instr_id']
            assert all(c['instr_id'] == instr_id for c in instance_candidates)
            if instr_id in candidate_lists_by_instr_id:
                looped = True
            else:
                candidate_lists_by_instr_id[instr_id] = instance_candidates
        if looped:
            break

    accuracies_by_weight, index_counts_by_weight = {}, {}
    for weight in speaker_weights:
        results, index_count = rational_mix(candidate_lists_by_instr_id, weight)
        if evaluator is not None:
            accuracies_by_weight[weight] = evaluator.score_results(results)[0]
        else:
            accuracies_by_weight[weight] = results
        index_counts_by_weight[weight] = index_count
    return accuracies_by_weight, index_counts_by_weight
