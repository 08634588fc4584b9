"""Follower rollout on the HIP path.

* `batch_instructions_from_encoded` -- host-side instruction batching with the reference's
  semantics (follower.py:75-105).
* `DeviceFollowerBatch` / `FollowerEngine` -- the whole follower episode
  (Seq2SeqAgent._rollout_with_loss, follower.py:430-539, and
  _score_obs_actions_and_instructions, :342-428) as a sync-free sequence of C-ABI calls over
  index-form observations: encoder, then per step decoder + glue, with the argmax /
  teacher feedback, `ended` flags, u_prev gather and loss terms all kept on the device.
  Training: `engine.rollout(...).loss.backward()` runs BPTT through the C-ABI backward
  entry points, accumulating weight gradients in place.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import call
from .features import cand_sincos
from .model import (decoder_params, decoder_w_struct, decoder_fold, _encoder_structs, _TAPE_KEYS,
                    grad_ptr, trainable_embedding)
from .runtime import ptr, stream, ws_args, wgrad_ws_args, ensure_workspace, dropout_arg, fill_regions, take_fault, PersistentLaunchFault, concurrent_stream, graph_capture, WeightsMoved
from .dp import collectives_on

byref = C.byref

PAD, UNK, EOS, BOS = 0, 1, 2, 3      # utils.py:19-24
FEEDBACK = {'teacher': 0, 'argmax': 1, 'sample': 2}


def batch_instructions_from_encoded(encoded_instructions, max_length, reverse=False, sort=False,
                                    device=None):
    """follower.py:75-105.  Returns (seq [B,max_length] int64, mask [B,max(len)] bool
    (True = PAD), lengths list[, perm list]); tensors on `device` (default: cuda if present)."""
    # (rows that are the SAME object -- the candidate routes of one instruction in the pragmatic re-ranking,
    # rational_follower.py:67-69 -- are encoded once)
    first, src = {}, []
    for inst in encoded_instructions:
        j = first.get(id(inst))
        if j is None:
            j = first[id(inst)] = len(first)
        src.append(j)
    distinct = [None] * len(first)
    for inst, j in zip(encoded_instructions, src):
        distinct[j] = inst
    seq = np.full((len(distinct), max_length), PAD, np.int64)
    lengths = []
    for i, inst in enumerate(distinct):
        inst = np.asarray(inst, np.int64)
        if len(inst) > 0:
            assert inst[-1] != EOS
        if reverse:
            inst = inst[::-1]
        inst = np.concatenate((inst, [EOS]))[:max_length]
        seq[i, :len(inst)] = inst
        lengths.append(len(inst))
    if len(distinct) < len(src):
        seq = seq[src]
        lengths = [lengths[j] for j in src]
    perm = None
    if sort:
        perm = np.argsort(-np.asarray(lengths), kind='stable')
        lengths = [lengths[i] for i in perm]
        seq = seq[perm]
    mask = (seq == PAD)[:, :max(lengths)]
    if device is None:
        device = 'cuda' if torch.cuda.is_available() else 'cpu'
    seq_t = torch.from_numpy(seq).to(device)
    mask_t = torch.from_numpy(np.ascontiguousarray(mask)).to(device)
    if sort:
        return seq_t, mask_t, lengths, list(perm)
    return seq_t, mask_t, lengths


@dataclass
class DeviceFollowerBatch:
    """Index-form episode batch resident in HBM (see synth.FollowerBatch for the fields)."""
    seq: torch.Tensor          # [B,Lpad] int64
    lengths: list
    lengths_dev: torch.Tensor  # [B] int32
    mask: torch.Tensor         # [B,T] uint8, 1 = PAD
    vp: torch.Tensor           # [S,B] int32
    view: torch.Tensor         # [S,B] int32
    a_num: torch.Tensor        # [S,B] int32
    cand_view: torch.Tensor    # [S,B,A] int32
    sincos: torch.Tensor       # [S,B,A,4] fp32
    target: torch.Tensor       # [S,B] int64
    a_max: int
    row0: int = 0              # global id of row 0 (data-parallel shard offset)

    @property
    def batch_size(self):
        return self.seq.shape[0]

    @classmethod
    def from_synth(cls, fb, device='cuda', max_length=80, reverse=True, rows=None, row0=0):
        """Upload a synth.FollowerBatch (optionally only `rows`, a slice, for data parallelism)."""
        sl = rows if rows is not None else slice(None)
        instr = fb.instr[sl]
        seq, mask, lengths = batch_instructions_from_encoded(instr, max_length, reverse=reverse,
                                                             device=device)
        dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)  # noqa: E731
        return cls(seq=seq, lengths=lengths,
                   lengths_dev=torch.tensor(lengths, dtype=torch.int32, device=device),
                   mask=mask.to(torch.uint8).contiguous(),
                   vp=dev(fb.vp[:, sl], torch.int32), view=dev(fb.view[:, sl], torch.int32),
                   a_num=dev(fb.a_num[:, sl], torch.int32),
                   cand_view=dev(fb.cand_view[:, sl], torch.int32),
                   sincos=dev(cand_sincos(fb.cand_heading[:, sl], fb.cand_elevation[:, sl]),
                              torch.float32),
                   target=dev(fb.target[:, sl], torch.int64), a_max=fb.a_max, row0=row0)


class RolloutState:
    """Everything one rollout keeps in HBM: per-step tapes and the glue outputs."""
    pass


class _RolloutLossFn(torch.autograd.Function):
    """Connects the engine's device-side loss to torch autograd: backward() runs BPTT."""

    @staticmethod
    def forward(ctx, engine, state, *params):
        ctx.engine, ctx.state = engine, state
        return state.loss_buf.clone().reshape(())

    @staticmethod
    def backward(ctx, dloss):
        if ctx.state is None:
            raise RuntimeError('second backward through a pass whose tape the first backward released')
        ctx.engine._backward(ctx.state, dloss)
        # state -> loss -> grad_fn -> ctx -> state is a reference cycle: a few hundred MB of tape per pass that only
        # the cyclic collector would free (seen as a 4 GB sawtooth under a training loop).  The tape has been consumed.
        ctx.state = None
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class FollowerEngine:
    def __init__(self, encoder, decoder, store, group=None):
        self.encoder, self.decoder, self.store = encoder, decoder, store
        self.group = group              # torch.distributed process group for data parallelism
        self.iteration = 0              # rollouts issued so far
        self.site_next = 0              # first unused dropout / sampling site (see rollout)
        # device-side site counter (capture_training): when set, the kernels receive stream ids RELATIVE to this word
        # and `site_next` only mirrors it on the host
        self.site_word = None
        self.dropout_seed = None
        self.two_stream_backward = True  # heads of the backward on a side stream (see sf_follower_episode_bwd)
        self._side_stream = None
        # BPTT in this many chunks with the weight-gradient products of finished chunks on a third stream.
        # Correct (tests/test_gpu_follower.py) and SLOWER on MI355X: 5.25 ms per iteration with one product
        # over all S*B rows at the end, 5.49 with 2 or 4 chunks, 5.7 with 10 (tools/train_time.py) -- a
        # chip-filling product does not hide beside the dependent chain, it delays it.  Off (1).
        self.wgrad_chunks = 1
        # the eight small weight-gradient products on a third stream beside the two LSTM ones: measured no gain
        # (5.29 vs 5.26 ms per iteration): off
        self.split_wgrad_streams = False
        self.encoder_backward_first = os.environ.get('SF_ENC_BWD_FIRST', '0') == '1'   # (experiment switch, see _backward)
        self.grad_sync = None            # dp.BucketedGrads(dp.follower_buckets(enc, dec)): all-reduce launched from the backward
        # the fold's two products on a second stream beside step 0's attention: built, correct, SLOWER -- the fork and the
        # join inside the replayed graph cost more than the 25 us they hide (1.813 against 1.759 ms per rollout): off
        self.fold_build_overlap = False
        self._open_forks = []            # side streams forked from the current one and not joined yet (_backward)
        # ... with q' = M_v h1 + c_v and [r | c] = M_a h~ + c_a as single products (ABI 9 chain_fold): three launches behind
        # the cell instead of four -- BUILT, CORRECT, SLOWER (round 6: 2.03 ms per rollout against 1.77: the [100 x 512] x
        # [2176 x 512]^T products re-read their A operand per 16-column block, 18.5 + 27.9 us for the two stages that replace
        # 10.7 + 7.5 + 14.4): off
        self.fold_chain = False
        self.fold_text = True            # inference rollouts: text attention over ctx W_in / ctx W_out[:, :H]^T (ABI 9)
        self._wgrad_stream = None
        # model.decoder_fold for no-grad eval rollouts (folded Linears + the folded paired schedule of
        # sf_attn_decoder_tail_fwd).  Correct (tests/test_gpu_follower.py) but SLOWER on MI355X: the two
        # folded [2176 x 512] products read 4.4 MB of weights each and take 11.6 us, the four unfolded
        # ones 5-6 us each in paired stages: 2.20 vs 2.09 ms per rollout.  Off.
        self.fold_inference = False
        self.pipelined = True           # head(t+1) next to tail(t) in paired launches (sf_hip.h)
        self.two_stream_forward = False  # experiment: visual half of step t+1 on a side stream, ordered by device flags
        self.episode_call = True        # the whole decode loop (and its backward) as ONE C call
        self.fused_env_step = True      # nav.DeviceNavBatch: the env step inside the scoring + glue launch
        self.fallbacks = 0              # rollouts re-issued on the per-step kernels after a persistent-launch fault (run)

    # ------------------------------------------------------------------------------ forward
    def rollout(self, batch, steps, feedback='argmax', train=None, finalize=True):
        """Runs encoder + `steps` decode steps.  Returns a RolloutState with
        .logits [S,B,A] (masked), .actions [S,B], .step_scores [S,B], .loss (0-dim tensor,
        differentiable when parameters require grad and grad mode is on), .h, .c, .ctx."""
        enc, dec, store = self.encoder, self.decoder, self.store
        dev = store.device
        B, A, S = batch.batch_size, batch.a_max, steps
        bidir = enc.num_directions == 2          # train.py:197-199: hidden_size // 2 per direction
        H, E = enc.hidden_size * enc.num_directions, enc.embedding_size
        F, V = store.F, store.V
        D = dec.visual_attention_layer.linear_in_h.weight.shape[0]
        T = batch.mask.shape[1]                  # (= max(batch.lengths), or wider: a batch of fixed shapes, nav.DeviceNavBatch)
        Lpad = batch.seq.shape[1]
        training = dec.training if train is None else train
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        st = RolloutState()
        st.batch, st.steps, st.dims = batch, S, (B, A, H, E, F, V, D, T)
        st.feedback = FEEDBACK[feedback]

        # dropout configuration: one seed per engine, site ids advance with the iteration
        p_dec = dec.drop.p if training else 0.0
        p_enc = enc.drop.p if training else 0.0
        if self.dropout_seed is None:
            self.dropout_seed = torch.initial_seed() & 0xFFFFFFFF
        st.drop_dec = (p_dec, self.dropout_seed, batch.row0)
        st.drop_enc = (p_enc, self.dropout_seed ^ 0x5BD1E995, batch.row0)
        # A rollout of S steps uses sites site0 .. site0 + S + 1 (x2 for the two masks of a step): the
        # next rollout starts behind them, at least 64 further on (so that S <= 62 keeps the
        # `iteration * 64` numbering the oracle tests mirror).  Identical on every data-parallel rank.
        st.site0 = self.site_next
        st.site_stride = max(64, S + 2)
        self.site_next += st.site_stride
        self.iteration += 1
        # what the kernels are given: the absolute site, or (device-side counter) 0 + the word's address
        st.site_dev = C.c_void_p(self.site_word.data_ptr()) if self.site_word is not None else None
        st.site_rel = 0 if self.site_word is not None else st.site0
        st.drop_dec += (st.site_dev, 1)
        st.drop_enc += (st.site_dev, 1)

        # ---- encoder (model.py:81-104)
        st.ctx = new(B, T, H)
        st.hs_all = new(steps + 1, B, H)
        st.cs_all = new(steps + 1, B, H)
        st.h_init, st.c_init = st.hs_all[0], st.cs_all[0]
        # (an inference rollout keeps no embedded tokens / gate tape: nothing will run backward)
        keep = training or (torch.is_grad_enabled() and any(
            p.requires_grad for p in list(enc.parameters()) + list(dec.parameters())))
        st.enc_tape = {} if bidir else \
            dict(emb=new(T, B, E) if keep else None, xg=new(T, B, 4 * H) if keep else None,
                 gates=new(T, B, 4 * H) if keep else None, hs=new(T + 1, B, H), cs=new(T + 1, B, H))
        etp = None if bidir else _lib.EncoderTape(*(st.enc_tape[k].data_ptr() if st.enc_tape[k] is not None else None
                                 for k in ('emb', 'xg', 'gates', 'hs', 'cs')))
        # a trainable (non-GloVe) embedding with a backward to follow: the input product is formed from the embedded
        # (train mode: dropped, model.py:86-87) tokens, not read from the cached table
        st.enc_table = not (keep and trainable_embedding(enc))
        st.enc_graph = None
        if bidir:
            # the module's composition of the two directions (model.EncoderLSTM._forward_bidirectional); its autograd
            # graph is the encoder's tape, and _backward() enters it with the decoder's (dctx, dh, dc)
            p_e, seed_e, row0 = st.drop_enc[:3]
            cfg = (p_e, (seed_e + 0x9E3779B1 * row0) & 0xFFFFFFFF, st.site0)
            with torch.set_grad_enabled(keep and torch.is_grad_enabled()):
                ctx_e, h_e, c_e = enc._forward_bidirectional(batch.seq, batch.lengths_dev, T, cfg, st.enc_table)
            st.ctx = ctx_e.detach()
            st.h_init.copy_(h_e.detach())
            st.c_init.copy_(c_e.detach())
            if ctx_e.requires_grad:
                st.enc_graph = (ctx_e, h_e, c_e)
        else:
            ew = _encoder_structs(enc, table=st.enc_table)
            call('sf_encoder_lstm_fwd', byref(ew), B, Lpad, T, E, H, ptr(batch.seq),
                 ptr(batch.lengths_dev), ptr(st.ctx), ptr(st.h_init), ptr(st.c_init), byref(etp),
                 dropout_arg(*st.drop_enc), st.site_rel, *ws_args(dev))

        # ---- decode steps
        shapes = dict(t_v=(D,), q=(F,), alpha_v=(V,), xin=(2 * F,), gates=(4 * H,), c1=(H,),
                      h1=(H,), cat2=(2 * H,), t_text=(H,), alpha=(T,), h_tilde=(H,), t_a=(D,),
                      wt=(D,), r=(F,), logit=(A,))
        # one extra xin slot: step t's glue writes dropout(u_next) straight into step t+1's LSTM input
        shapes_x = dict(shapes, xin=(2 * F,))
        st.tape = {k: new(S + (1 if k == 'xin' else 0), B, *shapes_x[k]) for k in _TAPE_KEYS
                   if k not in ('h1', 'c1')}
        # hidden / cell states of all steps stacked: hs[t] is step t's h0, hs[t+1] its h1 -- the
        # batched weight-gradient products read hs[0:S] as one [S*B, H] matrix
        st.hs, st.cs = st.hs_all, st.cs_all
        st.tape['h1'], st.tape['c1'] = st.hs[1:], st.cs[1:]
        st.ended = torch.empty(B, dtype=torch.uint8, device=dev)
        # u_begin = 0 (model.py:368; the feature half of the row is written by the attention), no row has ended
        # (follower.py:380): one launch
        fill_regions((st.tape['xin'][0], 0.0), (st.ended, 0))
        st.actions = torch.empty(S, B, dtype=torch.int64, device=dev)
        st.target_used = torch.empty(S, B, dtype=torch.int64, device=dev)
        st.step_scores, st.ce_term, st.live = new(S, B), new(S, B), new(S, B)
        st.sum_cnt, st.gscale, st.loss_buf = new(S, 2), new(S), new(1)
        params = decoder_params(dec)
        all_params = list(params) + [p for p in enc.parameters()]
        st.differentiable = torch.is_grad_enabled() and any(p.requires_grad for p in all_params)
        # inference: consecutive Linears folded (two dependent stages fewer per step); the backward
        # needs the unfolded intermediates, so a differentiable rollout keeps them
        fold = None if (st.differentiable or training or not self.fold_inference) else decoder_fold(dec)
        dw = decoder_w_struct(params, fold=fold)
        ws = ws_args(dev)
        d_dec = _lib.Dropout(float(st.drop_dec[0]), int(st.drop_dec[1]) & 0xFFFFFFFF, int(st.drop_dec[2]), st.site_dev, 1)
        d_ptr = C.pointer(d_dec) if st.drop_dec[0] else None
        # A nav.DeviceNavBatch produces its observations ON THE DEVICE, one step ahead of the decoder,
        # from the action the glue kernel has just chosen (real student forcing: the next panorama
        # depends on a_t).  The attention of step t+1 therefore cannot ride beside tail(t); its QUERY can
        # (t_v', q' need only h1): tail(t) with tape_next but no X_next, env step, then the attention alone.
        on_device_env = hasattr(batch, 'advance')
        if on_device_env:
            batch.advance(-1)                           # slot 0 = the initial observation
        pipelined = self.pipelined and not on_device_env
        st.episode = None
        # the same one-call episode for a device-resident environment (the env step inside every scoring + glue launch,
        # the attention of step t + 1 behind it): no host work between the launches of a TRAINING rollout either, and
        # the backward takes the two-stream episode path
        nav_episode = (on_device_env and self.pipelined and self.episode_call and self.fused_env_step and fold is None
                       and bool(dw.visual.w_v_t))
        if (pipelined or nav_episode) and self.episode_call:
            # every per-step tensor is a stacked [S][...] array: hand step 0 to the library once
            ep = _lib.FollowerEpisode()
            ep.S, ep.B, ep.H, ep.D, ep.L, ep.A = S, B, H, D, T, A
            ep.X = store.pano(batch.vp[0], batch.view[0])
            ep.U = store.cands(batch.vp[0], batch.cand_view[0], batch.sincos[0], batch.a_num[0], A)
            ep.h_init, ep.c_init = st.h_init.data_ptr(), st.c_init.data_ptr()
            ep.ctx, ep.ctx_mask = st.ctx.data_ptr(), batch.mask.data_ptr()
            ep.tape = _lib.DecoderTape(*(st.tape[k][0].data_ptr() for k in _TAPE_KEYS))
            ep.glue = _lib.FollowerGlue(
                None, batch.target[0].data_ptr(), st.feedback, st.ended.data_ptr(),
                st.actions[0].data_ptr(), st.target_used[0].data_ptr(), st.step_scores[0].data_ptr(),
                None, 0, None, 0, st.ce_term[0].data_ptr(), st.live[0].data_ptr(),
                int(st.drop_dec[1]) ^ 0x1B873593, 0, batch.row0, st.site_dev)
            ep.drop = d_dec
            ep.step0 = st.site_rel
            ep.side_stream = None
            if nav_episode:
                st.navio0 = batch.fused_step(0)               # (sf_nav_io of step 0; the library strides it per step)
                ep.glue.nav = C.cast(C.pointer(st.navio0), C.c_void_p)
            if self.two_stream_forward and fold is None and not nav_episode:
                if self._side_stream is None:
                    self._side_stream = concurrent_stream(dev)
                ep.side_stream = self._side_stream.cuda_stream
            # INFERENCE: the text attention in folded form (include/sf_hip.h, ABI 9: ctx_q / ctx_o; csrc/sf_attention.hip:
            # text_fold_body) -- the context is constant over the episode, so W_in and W_out[:, :H] are applied to it ONCE
            # and two dependent launches leave every decode step.  Nothing is taped for a backward, hence never for a
            # differentiable or train-mode rollout; one stream only.
            st.text_folded = (self.fold_text and not st.differentiable and not training and fold is None
                              and ep.side_stream is None and S > 1 and T <= 80 and not bidir)
            if st.text_folded and self.fold_build_overlap:
                # the two fold products (they need the encoder's context only) on a second stream beside step 0's attention
                if self._side_stream is None:
                    self._side_stream = concurrent_stream(dev)
                    ensure_workspace(self._side_stream, dev)
                ep.side_stream = self._side_stream.cuda_stream
            if st.text_folded:
                st.ctx_fold = new(2, B, T, H)
                ep.ctx_q, ep.ctx_o = st.ctx_fold[0].data_ptr(), st.ctx_fold[1].data_ptr()
                if self.fold_chain:
                    # ... and the folded query / scoring matrices (model.decoder_fold: rebuilt in place per weight
                    # version): three dependent launches behind the cell instead of four
                    st.chain_fold = decoder_fold(dec)
                    ep.chain_fold = C.cast(C.pointer(st.chain_fold), C.c_void_p)
            call('sf_follower_episode_fwd', byref(dw), byref(ep), *ws)
            st.episode = (ep, dw)
        tapes = [] if st.episode else [_lib.DecoderTape(*(st.tape[k][t].data_ptr() for k in _TAPE_KEYS))
                                       for t in range(S)]
        panos = [] if st.episode else [store.pano(batch.vp[t], batch.view[t]) for t in range(S)]
        if pipelined and not st.episode:
            call('sf_attn_decoder_head_fwd', byref(dw), byref(panos[0]), B, H, D, ptr(st.h_init),
                 byref(tapes[0]), d_ptr, st.site_rel, *ws)
        deferred = on_device_env and self.pipelined and fold is None and dw.visual.w_v_t and not st.episode
        if deferred:
            call('sf_attn_decoder_head_fwd', byref(dw), byref(panos[0]), B, H, D, ptr(st.h_init),
                 byref(tapes[0]), d_ptr, st.site_rel, *ws)
        for t in range(0 if not st.episode else S, S):
            pano = panos[t]
            cnd = store.cands(batch.vp[t], batch.cand_view[t], batch.sincos[t], batch.a_num[t], A)
            tp = tapes[t]
            h0 = st.h_init if t == 0 else st.tape['h1'][t - 1]
            c0 = st.c_init if t == 0 else st.tape['c1'][t - 1]
            glue = _lib.FollowerGlue(
                None, batch.target[t].data_ptr(), st.feedback, st.ended.data_ptr(),
                st.actions[t].data_ptr(), st.target_used[t].data_ptr(), st.step_scores[t].data_ptr(),
                st.tape['xin'][t + 1].data_ptr(), 2 * F, d_ptr, 2 * (st.site_rel + t + 1),
                st.ce_term[t].data_ptr(), st.live[t].data_ptr(),
                int(st.drop_dec[1]) ^ 0x1B873593, st.site_rel + t, batch.row0, st.site_dev)
            navio = None
            if deferred and self.fused_env_step:        # env.step + observe in the scoring + glue launch
                navio = batch.fused_step(t)
                glue.nav = C.cast(C.pointer(navio), C.c_void_p)
            if pipelined or deferred:
                nxt = t + 1 < S
                call('sf_attn_decoder_tail_fwd', byref(dw), byref(cnd), B, H, D, T, None, ptr(h0),
                     ptr(c0), ptr(st.ctx), ptr(batch.mask), None, byref(tp), byref(glue), d_ptr,
                     st.site_rel + t, byref(panos[t + 1]) if nxt and not deferred else None,
                     byref(tapes[t + 1]) if nxt else None, *ws)
            else:
                call('sf_attn_decoder_fwd', byref(dw), byref(pano), byref(cnd), B, H, D, T, None,
                     ptr(h0), ptr(c0), ptr(st.ctx), ptr(batch.mask), None, byref(tp), byref(glue),
                     d_ptr, st.site_rel + t, *ws)
            if on_device_env:                           # env.step + observe + teacher for step t + 1
                if navio is None:
                    batch.advance(t, st.actions[t], st.ended)
                if deferred and t + 1 < S:              # the attention of step t + 1 over the panorama just chosen
                    call('sf_attn_decoder_attend_fwd', byref(panos[t + 1]), B, byref(tapes[t + 1]), d_ptr,
                         st.site_rel + t + 1, *ws)
        call('sf_reduce_terms', ptr(st.ce_term), ptr(st.live), S, B, ptr(st.sum_cnt), ws[2])
        st.logits = st.tape['logit']
        st.h, st.c = st.tape['h1'][S - 1], st.tape['c1'][S - 1]
        st.all_params = all_params
        if not finalize:            # a row shard of a larger batch: the caller combines sum_cnt tables
            return st               # and calls finish(st, total)
        if self.group is not None:
            # global per-step normaliser so that the sharded loss equals the reference's batch mean
            # (a collective point: under a segmented capture the graph is cut here, runtime.TrainingGraph)
            table, grp = st.sum_cnt, self.group
            self._collective(lambda: torch.distributed.all_reduce(table, group=grp))
        return self.finish(st)

    collective_hook = None        # runtime.TrainingGraph (segmented): takes the host action of a collective point

    def _collective(self, action):
        if self.collective_hook is not None:
            self.collective_hook(action)
        else:
            action()

    def run(self, batch, steps, feedback='argmax', train=None, backward=False, while_running=None):
        """`rollout` (and, with backward=True, `loss.backward()`) + the fault check of the persistent encoder
        launches (include/sf_hip.h: sf_workspace_fault_offset): ONE host sync; a launch that gave up a bounded wait
        (co-residency lost to another process) has poisoned its outputs with NaN, so the SAME rollout -- same
        dropout / sampling sites -- is re-issued in this process with the per-step encoder kernels
        (SF_ENC_PER_STEP), after zeroing the gradients the poisoned backward accumulated.  Under a process group
        the decision is taken on the MAX of all ranks' fault words, so every rank re-issues together; gradient
        buckets the poisoned backward had launched are waited for and re-armed (`BucketedGrads.abort`) first.  Raises PersistentLaunchFault if a fault is still raised afterwards.
        `while_running()`: host work the caller wants done between the issue and the sync (the next minibatch)."""
        dev = self.store.device
        site, it = self.site_next, self.iteration

        def once():
            st = self.rollout(batch, steps, feedback, train)
            if backward:
                st.loss.backward()
            nonlocal while_running
            if while_running is not None:
                while_running, cb = None, while_running
                cb()
            bits = take_fault(dev)
            if self.group is not None and collectives_on(self.group):
                # (MAX, not BOR: RCCL has no bitwise reductions; the caller only needs "some rank faulted")
                t = torch.tensor([bits], device=dev, dtype=torch.int32)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.group)
                bits = int(t.item())
            return st, bits

        st, bits = once()
        if bits:
            self.fallbacks += 1
            if backward:
                # the poisoned backward has already launched its gradient buckets (on every rank alike): let them
                # land before the buffer is zeroed, and re-arm the buckets for the re-issued backward
                if self.grad_sync is not None:
                    self.grad_sync.abort()
                for p_ in st.all_params:
                    if p_.grad is not None:
                        p_.grad.zero_()
            keep = getattr(self.encoder, 'persistent', True)
            self.encoder.persistent, self.site_next, self.iteration = False, site, it
            try:
                st, again = once()
            finally:
                self.encoder.persistent = keep
            if again:
                raise PersistentLaunchFault('fault bits %d, and %d after the per-step re-issue' % (bits, again))
        return st

    def finish(self, st, total=None):
        """Second half of `rollout(..., finalize=False)`: turns the per-step (CE sum, live count) table
        into the loss.  `total` = the table summed over every row shard of the batch (what the
        data-parallel all-reduce produces across GPUs); with it the shard's loss IS the whole batch's
        loss (follower.py:481: per-step mean over the non-ignored rows of the full batch) and its
        backward() contributes exactly this shard's share of the batch gradient."""
        if total is not None:
            st.sum_cnt = total
        call('sf_loss_finalize', ptr(st.sum_cnt), st.steps, ptr(st.loss_buf), ptr(st.gscale), stream())
        if st.differentiable:
            st.loss = _RolloutLossFn.apply(self, st, *st.all_params)
        else:
            st.loss = st.loss_buf.reshape(())
        return st

    def _baked_pointers(self):
        """Every weight-side device pointer a captured rollout bakes into its hipGraph: the parameters
        and their derived copies (transposed layouts, the encoder's [vocab,4H] table).  Building the
        structs also refreshes stale derived copies IN PLACE, on the current stream."""
        enc = self.encoder
        if enc.num_directions == 2:
            e2d = enc.encoder2decoder
            ew = b''.join(bytes(_encoder_structs(enc, direction=d)) for d in (0, 1)) + \
                repr((e2d.weight.data_ptr(), e2d.bias.data_ptr())).encode()
        else:
            ew = bytes(_encoder_structs(enc))
        dw = decoder_w_struct(decoder_params(self.decoder))
        fold = bytes(decoder_fold(self.decoder)) if (self.fold_inference or (self.fold_text and self.fold_chain)) else b''
        return ew + bytes(dw) + fold

    def _guarded(self, graph_replay):
        baked = self._baked_pointers()

        def replay():
            # weights updated since capture (optimizer.step, load_state_dict)?  Their derived copies are
            # rebuilt in place here, ahead of the replay on the same stream, so the graph reads current
            # data everywhere.  A MOVED tensor cannot be patched into the graph: refuse.
            if self._baked_pointers() != baked:
                raise WeightsMoved('a weight (or one of its cached layouts) moved since this rollout was '
                                   'captured; capture() again')
            graph_replay()
        return replay

    def capture(self, batch, steps, feedback='argmax'):
        """hipGraph of one inference rollout (eval mode, no autograd): returns (replay, state).
        `replay()` re-runs encoder + `steps` decode steps on the captured buffers; the state's
        tensors (.actions, .logits, .loss_buf, ...) are overwritten by every replay.

        The graph holds raw device pointers of the weights AND of their cached derived copies
        (`runtime.transposed`, the encoder's embedding x W_ih^T table).  Those copies live in
        persistent buffers that are refreshed in place, and `replay()` refreshes them first when a
        weight's version changed, so replays after `optimizer.step()` / `load_state_dict` use the new
        weights; if a weight tensor itself was re-allocated, `replay()` raises and the rollout must be
        captured again.  Dropout sites are fixed at capture time: a captured TRAINING rollout would
        replay one mask forever, which is why only inference is captured here."""
        # `sample` feedback: the sampling stream of a replay is a device word replay() writes first (kernel arguments are
        # frozen in a graph): every replay draws new actions (sf_follower_glue.sample_site_dev)
        sampled = feedback == 'sample'
        ctl = torch.zeros(4, dtype=torch.int32, device=self.store.device) if sampled else None
        with torch.no_grad():
            self.rollout(batch, steps, feedback, train=False)          # warm-up: allocations, caches
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            ensure_workspace(side, self.store.device)                  # (created and zero-filled OUTSIDE the graph)
            side.wait_stream(torch.cuda.current_stream())
            keep = (self.site_next, self.iteration)
            self.site_word = ctl[0:1] if sampled else None
            try:
                with torch.cuda.stream(side):
                    with graph_capture(graph, side):
                        st = self.rollout(batch, steps, feedback, train=False)
            finally:
                self.site_word = None
            if sampled:
                self.site_next, self.iteration = keep                  # (the capture ran nothing)
            torch.cuda.current_stream().wait_stream(side)
        # (the graph bakes the capture stream's workspace -- runtime.workspace is keyed on the stream handle -- so the
        # stream lives as long as the state: a recycled handle would hand the same scratch to someone else)
        st.capture_stream = side

        def graph_replay():
            if sampled:
                call('sf_store_u32x4', C.c_void_p(ctl.data_ptr()), int(self.site_next) & 0xFFFFFFFF, 0, 0, 0, stream())
                st.site0 = self.site_next
                self.site_next += st.site_stride
                self.iteration += 1
            graph.replay()
        return self._guarded(graph_replay), st

    def capture_training(self, batch, steps, feedback='sample', optimizers=(), zero=None):
        """hipGraph of ONE WHOLE TRAINING ITERATION (follower.py:1001-1020 + train.py:263-268): zero the gradients,
        rollout in train mode (dropout on, `feedback` as given), BPTT, `optimizer.step()` for every optimizer
        (optim.FusedAdam).  Returns a runtime.TrainingGraph; `.replay()` is one iteration, `.state` the rollout state
        its replays overwrite.  Runs one eager iteration first (a real training step).

        Every replay draws fresh dropout masks / samples and takes the next Adam step: sites and step counters are
        device words the graph reads (sf_dropout.site_dev, sf_adam_step_dev), numbered exactly like the eager loop's.
        Single process only (a gradient all-reduce cannot live in the graph); unidirectional encoder; pre-drawn
        observations or a device-resident environment (nav.DeviceNavBatch)."""
        from .runtime import TrainingGraph
        if self.encoder.num_directions == 2:
            raise NotImplementedError('capture_training: unidirectional encoder only')
        opts = list(optimizers)
        if self.group is not None or self.grad_sync is not None:
            return self._capture_training_dp(batch, steps, feedback, opts, zero)

        def body():
            if zero is not None:
                zero.zero()
            else:
                for o in opts:
                    o.zero_grad()
            st = self.rollout(batch, steps, feedback, train=True)
            st.loss.backward()
            for o in opts:
                o.step()
            return st
        return TrainingGraph(self, body, opts, self.store.device)

    def _capture_training_dp(self, batch, steps, feedback, opts, zero):
        """The DATA-PARALLEL iteration as replayed SEGMENTS (runtime.TrainingGraph, segmented): a collective cannot
        live in the graph (gloo stages through the host; RCCL's kernels belong to the process group's own stream), so the
        capture is CUT at every collective point -- the all-reduce of the per-step (CE sum, count) table behind the
        forward, the start of each gradient bucket's all-reduce behind the launches that complete it, the wait in front
        of the optimizers -- and a replay is: segment, host action, segment, ...: the launches of an iteration are still
        issued as a handful of graph launches instead of ~400 kernel launches, the collectives keep the order and the
        overlap of the eager loop (each is started behind exactly the work that precedes it on the replay stream; the
        persistent encoder backward runs in the segment BEHIND the 40 MB bucket's start).  Same sites, same Adam steps,
        same numbers as the eager data-parallel loop (tests/test_gpu_dataparallel.py)."""
        from .runtime import TrainingGraph
        sync = self.grad_sync
        if sync is None:
            raise ValueError('a data-parallel capture needs engine.grad_sync (dp.BucketedGrads: the buffer the '
                             'all-reduces work on)')
        zero = zero if zero is not None else sync
        dev = self.store.device

        def body():
            zero.zero()
            with torch.no_grad():                       # (the backward is called directly: every launch on this thread)
                st = self.rollout(batch, steps, feedback, train=True)
                self._backward(st, torch.ones((), device=dev))
            sync.wait()
            for o in opts:
                o.step()
            return st
        return TrainingGraph(self, body, opts, dev, segmented=(sync,))

    def capture_sharded(self, shards, steps, feedback='argmax'):
        """Runs the row shards of ONE batch as concurrent chains (inference): one hipGraph per
        shard, replayed on its own stream.  Samples never interact in the forward pass, and at batch
        100 every stage of the chain is latency-bound with most of the 256 CUs idle, so two
        half-batch chains overlap; the per-step (CE sum, live count) tables are added before the loss
        is finalised, exactly like the data-parallel path does across GPUs.  (One graph per stream:
        branches of a single hipGraph do run concurrently on ROCm 7.2 -- tools/graph_branch_probe.py --
        but every fork / join between them costs several microseconds.)  Returns (replay, states,
        loss_buf)."""
        dev = self.store.device
        streams = [torch.cuda.Stream() for _ in shards]
        graphs, states = [], []
        with torch.no_grad():
            for sh, s in zip(shards, streams):
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self.rollout(sh, steps, feedback, train=False)          # warm-up on this stream
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with graph_capture(g, s):
                        states.append(self.rollout(sh, steps, feedback, train=False, finalize=False))
                graphs.append(g)
            torch.cuda.synchronize()
        total = torch.empty(steps, 2, device=dev)
        loss_buf = torch.empty(1, device=dev)
        gscale = torch.empty(steps, device=dev)

        def replay_all():
            cur = torch.cuda.current_stream()
            for s, g in zip(streams, graphs):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    g.replay()
            for s in streams:
                cur.wait_stream(s)
            total.copy_(states[0].sum_cnt)
            for st in states[1:]:
                call('sf_add_f32', ptr(total), ptr(st.sum_cnt), total.numel(), stream())
            call('sf_loss_finalize', ptr(total), steps, ptr(loss_buf), ptr(gscale), stream())

        return self._guarded(replay_all), states, loss_buf

    # ------------------------------------------------------------------------------ backward
    def _backward(self, st, dloss):
        """BPTT.  Per step only the DATA gradients are formed (sequential dependency); every
        weight gradient is one product over all S*B stacked rows at the end (sf_attn_decoder_wgrad),
        accumulated in place into param.grad."""
        enc, dec, store = self.encoder, self.decoder, self.store
        batch, S = st.batch, st.steps
        B, A, H, E, F, V, D, T = st.dims
        dev = store.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        gscale = st.gscale * dloss.to(torch.float32)        # d loss / d step-loss, per step
        params = decoder_params(dec)
        dw, dg = decoder_w_struct(params), decoder_w_struct(params, grad=True)
        gshapes = dict(dgates=(4 * H,), dpre=(H,), dt_text=(H,), dt_v=(D,), dq=(F,), dwt=(D,),
                       dta=(D,), dr=(F,), dc=())
        gkeys = ('dgates', 'dpre', 'dt_text', 'dt_v', 'dq', 'dwt', 'dta', 'dr', 'dc')
        gt = {k: new(S, B, *gshapes[k]) for k in gkeys}
        dlogit = new(B, A)
        dh_a, dc_a, dh_b, dc_b = new(B, H), new(B, H), new(B, H), new(B, H)
        dctx = torch.zeros(B, T, H, device=dev, dtype=torch.float32)
        ws = ws_args(dev)
        d_dec = _lib.Dropout(float(st.drop_dec[0]), int(st.drop_dec[1]) & 0xFFFFFFFF, int(st.drop_dec[2]), st.site_dev, 1)
        d_ptr = C.pointer(d_dec) if st.drop_dec[0] else None
        dh1 = dc1 = None
        if st.episode is not None:
            ep, _ = st.episode
            # d[wc ; h1_drop] and d(score) of every step: the context gradient is formed once after
            # the loop instead of a read-modify-write of [B,T,H] per step
            gt_dcat2, gt_ds, gt_dh1d = new(S, B, 2 * H), new(S, B, T), new(S, B, H)
            gt0e = _lib.DecoderGTape(*(gt[k].data_ptr() for k in gkeys), gt_dcat2.data_ptr(), gt_ds.data_ptr(),
                                     gt_dh1d.data_ptr())
            # second stream: the scoring / text-attention backward of step t-1 runs beside the LSTM /
            # visual backward of step t (not under stream capture: eager issue only)
            # (under stream capture only with a side stream that already exists: probing one launches timed kernels)
            if self.two_stream_backward and (self._side_stream is not None or not torch.cuda.is_current_stream_capturing()):
                if self._side_stream is None:
                    self._side_stream = concurrent_stream(dev)
                ep.side_stream = self._side_stream.cuda_stream
            else:
                ep.side_stream = None
            # Backpropagation through time in chunks of steps: the weight-gradient products of a finished
            # chunk (matrix-core work over its stacked rows, accumulated in place) run on a third stream
            # while the earlier steps are still being walked (dependent, latency-bound launches that leave
            # most of the chip idle); only the last chunk's products are left for the end, beside the
            # encoder's backward.  `wgrad_chunks = 1` is the round-2 schedule (one product over all S*B rows).
            overlap_w = (self.wgrad_chunks > 1 and ep.side_stream is not None and S >= 2 * self.wgrad_chunks)
            n_chunks = self.wgrad_chunks if overlap_w else 1
            bounds = [(S * k) // n_chunks for k in range(n_chunks + 1)]
            which = C.c_int(0)
            tp_at = lambda t: _lib.DecoderTape(*(st.tape[k][t:].data_ptr() for k in _TAPE_KEYS))          # noqa: E731
            gt_at = lambda t: _lib.DecoderGTape(*(gt[k][t:].data_ptr() for k in gkeys), None, None, None)  # noqa: E731
            if overlap_w and self._wgrad_stream is None:
                self._wgrad_stream = concurrent_stream(dev, exclude=[x for x in (self._side_stream,) if x is not None])
            st.wgrad_done_from = S                   # steps >= this have their weight gradients issued
            for k in range(n_chunks - 1, -1, -1):
                lo, hi = bounds[k], bounds[k + 1]
                call('sf_follower_episode_bwd_range', byref(dw), byref(ep), byref(gt0e), ptr(gscale), ptr(dlogit),
                     ptr(dh_a), ptr(dc_a), ptr(dh_b), ptr(dc_b), ptr(dctx), byref(which), lo, hi,
                     ptr(dh1), ptr(dc1), *ws)
                dh1, dc1 = (dh_b, dc_b) if which.value else (dh_a, dc_a)
                if overlap_w and k > 0:
                    wst = self._wgrad_stream
                    wst.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(wst):
                        tpk, gtk = tp_at(lo), gt_at(lo)
                        call('sf_attn_decoder_wgrad', byref(dw), byref(dg), (hi - lo) * B, H, D, F, ptr(st.hs[lo:]),
                             byref(tpk), byref(gtk), *ws_args(dev))
                    st.wgrad_done_from = lo
        if getattr(self, '_bptt_only', False):          # (tools/bptt_overlap_probe.py times the backward through time alone)
            if self._side_stream is not None:
                torch.cuda.current_stream().wait_stream(self._side_stream)
            return
        for t in range(S - 1 if st.episode is None else -1, -1, -1):
            pano = store.pano(batch.vp[t], batch.view[t])
            cnd = store.cands(batch.vp[t], batch.cand_view[t], batch.sincos[t], batch.a_num[t], A)
            tp = _lib.DecoderTape(*(st.tape[k][t].data_ptr() for k in _TAPE_KEYS))
            gtp = _lib.DecoderGTape(*(gt[k][t].data_ptr() for k in gkeys), None, None, None)
            call('sf_follower_glue_bwd', B, A, ptr(st.tape['logit'][t]), ptr(st.target_used[t]),
                 ptr(gscale[t:t + 1]), ptr(dlogit), ws[2])
            call('sf_attn_decoder_bwd', byref(dw), None, byref(pano), byref(cnd), B, H, D, T,
                 ptr(st.hs[t]), ptr(st.cs[t]), ptr(st.ctx), byref(tp), byref(gtp), ptr(dlogit),
                 ptr(dh1), ptr(dc1), ptr(dh_a), ptr(dc_a), ptr(dctx), d_ptr, st.site_rel + t, *ws)
            dh1, dc1 = dh_a, dc_a
            dh_a, dc_a, dh_b, dc_b = dh_b, dc_b, dh_a, dc_a
        # data parallelism: dp.BucketedGrads laid out as dp.follower_buckets -- each bucket's all-reduce is
        # launched from here, behind the launches that complete it
        sync = self.grad_sync
        # rows whose weight gradients are still to be formed: all of them, or the first chunk's
        Sw = getattr(st, 'wgrad_done_from', S) if st.episode is not None else S
        tp0 = _lib.DecoderTape(*(st.tape[k].data_ptr() for k in _TAPE_KEYS))
        gt0 = _lib.DecoderGTape(*(gt[k].data_ptr() for k in gkeys), None, None, None)
        # the decoder's weight gradients (a few large products over the stacked rows: matrix-core
        # work) and the encoder's backward through time (80 dependent, latency-bound steps) are
        # independent: issued on two streams they overlap
        overlap = self.two_stream_backward and (self._side_stream is not None or not torch.cuda.is_current_stream_capturing())
        if overlap:
            side = self._wgrad_stream if (st.episode is not None and Sw < S) else self._side_stream
            if side is None:
                side = self._side_stream = concurrent_stream(dev)
            side.wait_stream(torch.cuda.current_stream())
            self._open_forks = [side]                    # (a segmented capture joins / re-opens these at its cuts)
            third = None
            if self.split_wgrad_streams:
                # the eight small products (25 TFLOP/s between them) beside the two LSTM ones (dW_ih alone fills the chip
                # at 0.76 of the matrix peak) instead of behind them: two streams, same accumulation targets
                if self._wgrad_stream is None:
                    self._wgrad_stream = concurrent_stream(dev, exclude=[x for x in (self._side_stream,) if x is not None])
                third = self._wgrad_stream if self._wgrad_stream is not side else self._side_stream
                third.wait_stream(torch.cuda.current_stream())
                self._open_forks.append(third)
        # (Measured in round 5 by wall clock: the encoder first, the two latency-bound pieces side by side, a third stream
        # and chunked weight gradients are all equal or slower than this order; what shortened the tail was fewer launches,
        # gemm_tn_group.  The many-row weight-gradient tiles -- 512 threads x 188 VGPRs, 96 KB LDS -- cannot sit beside the
        # persistent encoder backward's 256-VGPR workgroup on a CU.  NOTE: rocprofv3 --kernel-trace serialises the queues;
        # its timelines show no overlap at all and must not be read for concurrency, tools/bptt_overlap_probe.py.)
        if overlap and not self.encoder_backward_first:
            self._issue_wgrad(side, third, dw, dg, params, Sw * B, H, D, F, st, tp0, gt0, dev, sync)
        elif not overlap:
            self._decoder_wgrad(dw, dg, params, Sw * B, H, D, F, st, tp0, gt0, wgrad_ws_args(dev), sync)
        if enc.num_directions == 2:
            if st.enc_graph is not None:         # the two directions' tapes: entered with the decoder's gradients
                torch.autograd.backward(list(st.enc_graph), [dctx, dh1, dc1])
        else:
            etp = _lib.EncoderTape(*(st.enc_tape[k].data_ptr() for k in ('emb', 'xg', 'gates', 'hs', 'cs')))
            ew = _encoder_structs(enc, table=st.enc_table)
            eg = _encoder_structs(enc, grad=True, seq=None if st.enc_table else batch.seq)
            call('sf_encoder_lstm_bwd', byref(ew), byref(eg), B, T, E, H, ptr(batch.lengths_dev),
                 ptr(st.h_init), ptr(dctx), ptr(dh1), ptr(dc1), byref(etp), dropout_arg(*st.drop_enc),
                 st.site_rel, *ws)
        if sync is not None:
            sync.launch(2)                       # encoder gradients: complete behind sf_encoder_lstm_bwd
        if overlap and self.encoder_backward_first:
            self._issue_wgrad(side, third, dw, dg, params, Sw * B, H, D, F, st, tp0, gt0, dev, sync)
        if overlap:
            torch.cuda.current_stream().wait_stream(side)
            if third is not None:
                torch.cuda.current_stream().wait_stream(third)
            self._open_forks = []

    def _issue_wgrad(self, side, third, dw, dg, params, M, H, D, F, st, tp0, gt0, dev, sync):
        """The decoder's weight gradients on the side stream(s) (which already wait for the backward through time)."""
        with torch.cuda.stream(side):
            self._decoder_wgrad(dw, dg, params, M, H, D, F, st, tp0, gt0, wgrad_ws_args(dev), sync,
                                part='lstm' if third is not None else 'all')
        if third is not None:
            with torch.cuda.stream(third):
                self._decoder_wgrad(dw, dg, params, M, H, D, F, st, tp0, gt0, wgrad_ws_args(dev), sync, part='rest')

    @staticmethod
    def _decoder_wgrad(dw, dg, params, M, H, D, F, st, tp0, gt0, ws, sync, part='all'):
        """sf_attn_decoder_wgrad on the current stream: `part` = 'lstm' (the two LSTM products + bias sums: the
        40 MB gradient bucket), 'rest' (the other decoder weights) or 'all' (LSTM first).  With a gradient-bucket
        sync each part's all-reduce is launched behind it."""
        if sync is None and part == 'all':
            call('sf_attn_decoder_wgrad', byref(dw), byref(dg), M, H, D, F, ptr(st.hs), byref(tp0), byref(gt0), *ws)
            return
        g_lstm, g_rest = _lib.DecoderW(), _lib.DecoderW()
        g_lstm.lstm = dg.lstm
        g_rest.visual, g_rest.text, g_rest.action = dg.visual, dg.text, dg.action
        if part in ('all', 'lstm'):
            call('sf_attn_decoder_wgrad', byref(dw), byref(g_lstm), M, H, D, F, ptr(st.hs), byref(tp0), byref(gt0), *ws)
            if sync is not None:
                sync.launch(0)
        if part in ('all', 'rest'):
            call('sf_attn_decoder_wgrad', byref(dw), byref(g_rest), M, H, D, F, ptr(st.hs), byref(tp0), byref(gt0), *ws)
            if sync is not None:
                sync.launch(1)
