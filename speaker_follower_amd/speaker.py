"""Speaker scoring / decoding on the HIP path.

`SpeakerEngine.score` is Seq2SeqSpeaker._score_obs_actions_and_instructions
(speaker.py:123-202) over index-form paths: SpeakerEncoderLSTM (model.py:437-457: per path
step visual attention + LSTMCell, padded steps included, no length masking), then up to
`instruction_len` SpeakerDecoderLSTM steps (model.py:487-519) with the per-step glue
(log-softmax, NLL with PAD ignored, teacher / argmax next word, EOS bookkeeping;
speaker.py:163-191) -- all as C-ABI calls on one stream, no host sync per word.
`.loss.backward()` runs BPTT through the C-ABI backward entry points.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import call
from .features import cand_sincos
from .follower import batch_instructions_from_encoded, FEEDBACK, PAD, EOS, BOS
from .model import _grads, trainable_embedding
from .runtime import ptr, stream, ws_args, ensure_workspace, dropout_arg, fill_regions, visual_query_fold, struct_of, transposed, take_fault, PersistentLaunchFault, graph_capture, WeightsMoved

byref = C.byref


@dataclass
class DeviceSpeakerBatch:
    """Index-form speaker batch in HBM (see synth.SpeakerBatch)."""
    instr_seq: torch.Tensor     # [B,Lmax] int64 targets (not reversed, EOS appended, PAD after)
    path_mask: torch.Tensor     # [B,Tp] uint8, 1 = padded path step
    vp: torch.Tensor            # [Tp,B] int32, -1 on padded steps (zero panorama)
    view: torch.Tensor          # [Tp,B] int32
    act: torch.Tensor           # [Tp,B] int32: 1 = move (use act_view/sincos), 0 = stop / padded -> zero row
    act_view: torch.Tensor      # [Tp,B] int32
    act_sincos: torch.Tensor    # [Tp,B,4] fp32
    row0: int = 0

    @property
    def batch_size(self):
        return self.instr_seq.shape[0]

    @classmethod
    def from_synth(cls, sb, device='cuda', max_length=80, row0=0):
        Tp, B = sb.vp.shape
        steps = np.arange(Tp)[:, None]
        live = steps < sb.path_len[None, :]
        seq, _, _ = batch_instructions_from_encoded(sb.instr, max_length, device=device)
        dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)  # noqa: E731
        return cls(instr_seq=seq,
                   path_mask=dev((~live).T.astype(np.uint8), torch.uint8),
                   vp=dev(np.where(live, sb.vp, -1), torch.int32),
                   view=dev(sb.view, torch.int32),
                   act=dev((live & ~sb.act_is_stop).astype(np.int32), torch.int32),
                   act_view=dev(sb.act_view, torch.int32),
                   act_sincos=dev(cand_sincos(sb.act_heading, sb.act_elevation), torch.float32),
                   row0=row0)


class SpeakerState:
    pass


class _SpeakerLossFn(torch.autograd.Function):
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


_DEC_TAPE = ('emb', 'gates', 'c1', 'h1', 'cat2', 't_text', 'alpha', 'h_tilde', 'logit')


class _StepNllFn(torch.autograd.Function):
    """One word step's glue (speaker.py:163-191) for the module-stepped path: picks the next word, records score / NLL /
    liveness, returns the step's mean NLL over the non-PAD targets (0 when there is none); backward = gscale (softmax - onehot)."""

    @staticmethod
    def forward(ctx, logit, target, feedback, ended, w_next, score, smp):
        B, vocab = logit.shape
        nll, live = torch.empty(B, device=logit.device), torch.empty(B, device=logit.device)
        call('sf_speaker_glue_fwd', B, vocab, vocab, ptr(logit), ptr(target), feedback, PAD, EOS, ptr(ended), ptr(w_next),
             ptr(score), ptr(nll), ptr(live), smp, stream())
        n = live.sum()
        inv = torch.where(n > 0, 1.0 / n.clamp(min=1.0), torch.zeros_like(n))
        ctx.save_for_backward(logit, target, inv)
        return nll.sum() * inv

    @staticmethod
    def backward(ctx, dloss):
        logit, target, inv = ctx.saved_tensors
        B, vocab = logit.shape
        dlogit = torch.empty_like(logit)
        gs = (dloss * inv).reshape(1).to(torch.float32).contiguous()
        call('sf_speaker_glue_bwd', B, vocab, vocab, ptr(logit), ptr(target), PAD, ptr(gs), ptr(dlogit), stream())
        return dlogit, None, None, None, None, None, None


class SpeakerEngine:
    def __init__(self, encoder, decoder, store, group=None):
        self.encoder, self.decoder, self.store = encoder, decoder, store
        self.group = group
        self.iteration = 0
        self.site_next = 0              # first unused dropout site (see score)
        self.site_word = None           # device-side site counter while a training graph is captured (runtime.TrainingGraph)
        self.dropout_seed = None
        self.persistent = True          # persistent launches allowed: the inference word loop as ONE launch (sf_speaker_decode),
                                        # the recurrence of teacher-forced passes (sf_speaker_teacher_fwd / _bwd)
        self.stacked_wgrad = True       # backward: weight gradients as one product over all S*B rows (False: per step)
        # teacher-forced passes: the recurrence as one persistent launch + the attention / projection / glue of all S*B
        # rows at once (sf_speaker_teacher_fwd / _bwd); False: the word loop step by step (or sf_speaker_decode)
        self.teacher_batched = True
        self.fold_query = True          # inference: the path encoder's attention query through the float64 fold (runtime.visual_query_fold)
        self.fallbacks = 0              # passes re-issued on the per-step kernels after a persistent-launch fault (run)

    def capture(self, batch, steps, feedback='teacher'):
        """hipGraph of one inference scoring / decoding pass: returns (replay, state); the state's
        tensors are overwritten by every replay (same contract as FollowerEngine.capture)."""
        # `sample` feedback: the sampling stream must differ between replays, and kernel arguments are frozen in a graph --
        # the pass reads it from a device word that replay() writes first (sf_sample.stream_dev)
        sampled = feedback == 'sample'
        ctl = torch.zeros(4, dtype=torch.int32, device=self.store.device) if sampled else None
        with torch.no_grad():
            self.score(batch, steps, feedback, train=False)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            ensure_workspace(side, self.store.device)              # (created and zero-filled OUTSIDE the graph)
            side.wait_stream(torch.cuda.current_stream())
            keep = (self.site_next, self.iteration)
            self.site_word = ctl[0:1] if sampled else None
            try:
                with torch.cuda.stream(side):
                    with graph_capture(graph, side):
                        st = self.score(batch, steps, feedback, train=False)
            finally:
                self.site_word = None
            if sampled:
                self.site_next, self.iteration = keep           # (the capture ran nothing)
            torch.cuda.current_stream().wait_stream(side)
        baked = bytes(self.decoder._w_struct())

        def replay():
            # the [vocab,4H] input-product table is refreshed in place when the weights changed; a
            # re-allocated weight cannot be patched into the graph (FollowerEngine.capture)
            if bytes(self.decoder._w_struct()) != baked:
                raise WeightsMoved('a speaker weight moved since this pass was captured; capture() again')
            # (the encoder's derived layouts likewise: transposed copies and the float64 query fold are rebuilt IN PLACE
            # when a weight's version moved -- the graph reads the same buffers)
            ep_ = self.encoder._params8()
            transposed(ep_[2])
            transposed(ep_[0])
            if self.fold_query:
                visual_query_fold(ep_[0], ep_[1], ep_[2])
            if sampled:
                call('sf_store_u32x4', C.c_void_p(ctl.data_ptr()), int(self.site_next) & 0xFFFFFFFF, 0, 0, 0, stream())
                st.site0 = self.site_next
                self.site_next += st.site_stride
                self.iteration += 1
            graph.replay()
        # The graph bakes the capture stream's workspace (runtime.workspace is keyed on the stream handle): the stream
        # must live as long as the graph, or a later stream could be handed the same handle -- and with it the same
        # scratch, exchange buffers and tickets -- while this graph still replays on them.
        replay.capture_stream = side
        st.capture_stream = side
        return replay, st

    def _score_modules(self, batch, steps, feedback, train):
        """speaker.py:123-202 stepped through the MODULES' own forward (torch autograd over C-ABI operators): the path
        for decoder variants the fused word loops do not cover (`use_input_att_feed`, model.py:500-513).  One launch
        sequence per word, one host sync at the end; slower than the fused paths by design."""
        enc, dec, store = self.encoder, self.decoder, self.store
        dev = store.device
        B, S = batch.batch_size, steps
        Tp = batch.vp.shape[0]
        F = store.F
        training = dec.training if train is None else train
        was = (enc.training, dec.training)
        enc.train(training)
        dec.train(training)
        try:
            st = SpeakerState()
            st.batch, st.steps, st.feedback = batch, S, FEEDBACK[feedback]
            st.teacher_path = st.persistent = False
            st.site0 = self.site_next
            st.site_stride = max(256, S + 2, Tp + 2)
            self.site_next += st.site_stride
            self.iteration += 1
            # dense inputs of the module API: action embeddings [Tp] x [B,F], panoramas [Tp] x [B,V,F] (zero on padded steps)
            act = torch.empty(Tp, B, F, device=dev)
            call('sf_gather_path_actions', ptr(store.table), store.V, store.IMG, store.LOC, ptr(batch.vp), ptr(batch.act_view),
                 ptr(batch.act_sincos), ptr(batch.act), Tp * B, ptr(act), F, stream())
            live_step = (batch.vp >= 0)
            feats = [store.gather_panorama(batch.vp[t].clamp(min=0), batch.view[t]) * live_step[t].view(B, 1, 1).float()
                     for t in range(Tp)]
            ctx, h, c = enc([act[t] for t in range(Tp)], feats)
            st.ctx = ctx
            mask = batch.path_mask
            st.words = torch.empty(S + 1, B, dtype=torch.int64, device=dev)
            st.words[0] = BOS
            st.ended = torch.zeros(B, dtype=torch.uint8, device=dev)
            st.step_scores = torch.zeros(S, B, device=dev)
            targets = batch.instr_seq[:, :S].t().contiguous()
            seed = (self.dropout_seed if self.dropout_seed is not None else torch.initial_seed()) & 0xFFFFFFFF
            logits, terms, all_ended = [], [], []
            for t in range(S):
                h, c, alpha, logit = dec(st.words[t].view(-1, 1), h, c, ctx, mask)
                smp = byref(_lib.Sample((seed ^ 0x3C6EF372) & 0xFFFFFFFF, st.site0 + t, batch.row0)) if st.feedback == 2 else None
                terms.append(_StepNllFn.apply(logit.contiguous(), targets[t], st.feedback, st.ended, st.words[t + 1],
                                              st.step_scores[t], smp))
                all_ended.append(st.ended.min().to(torch.float32))       # 1 once every row has produced EOS
                logits.append(logit.detach())
            # the reference leaves its loop behind the first step at which every row has ended (speaker.py:196-197): the
            # steps up to and INCLUDING that one count (no host sync: the cut is a device-side weight vector)
            done = torch.stack(all_ended)
            keep = ((torch.cumsum(done, 0) - done) == 0).to(torch.float32)
            loss = (torch.stack(terms) * keep).sum()
            st.logits = torch.stack(logits)
            st.h, st.c = h, c
            st.loss = loss
            st.loss_buf = loss.detach().reshape(1)
            return st
        finally:
            enc.train(was[0])
            dec.train(was[1])

    def score(self, batch, steps, feedback='teacher', train=None):
        """Returns a SpeakerState: .words [S,B], .logits [S,B,vocab], .step_scores [S,B],
        .loss (differentiable), .ctx [B,Tp,H]."""
        if getattr(self.decoder, 'use_input_att_feed', False):
            return self._score_modules(batch, steps, feedback, train)
        enc, dec, store = self.encoder, self.decoder, self.store
        dev = store.device
        B, S = batch.batch_size, steps
        Tp = batch.vp.shape[0]
        H, F, V = enc.hidden_size, store.F, store.V
        D = enc.visual_attention_layer.linear_in_h.weight.shape[0]
        E, vocab = dec.vocab_embedding_size, dec.vocab_size
        ldv = (vocab + 3) & ~3
        training = dec.training if train is None else train
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        st = SpeakerState()
        st.batch, st.steps, st.dims = batch, S, (B, Tp, H, F, V, D, E, vocab, ldv)
        st.feedback = FEEDBACK[feedback]
        if self.dropout_seed is None:
            self.dropout_seed = torch.initial_seed() & 0xFFFFFFFF
        # sites site0 .. site0 + max(S, Tp) + 1 are used by this pass; the next one starts behind them
        st.site0 = self.site_next
        st.site_stride = max(256, S + 2, Tp + 2)
        self.site_next += st.site_stride
        self.iteration += 1
        # what the kernels are given: the absolute site, or (device-side counter) 0 + the word's address.  The encoder's
        # raw stream ids are numbered 2 * (site + t) (+ 1): its configuration scales the device offset by 2
        st.site_dev = C.c_void_p(self.site_word.data_ptr()) if self.site_word is not None else None
        st.site_rel = 0 if self.site_word is not None else st.site0
        st.drop_enc = (enc.drop.p if training else 0.0, self.dropout_seed ^ 0x2545F491, batch.row0, st.site_dev, 2)
        st.drop_dec = (dec.drop.p if training else 0.0, self.dropout_seed, batch.row0, st.site_dev, 1)
        ws = ws_args(dev)
        # `sample` feedback (speaker.py:170-174): word step t draws from stream site0 + t of this seed, keyed on the
        # global row id (sf_sampling.h)
        smp = _lib.Sample((self.dropout_seed ^ 0x3C6EF372) & 0xFFFFFFFF, st.site_rel, batch.row0, st.site_dev)
        if self.group is not None and st.feedback != 0:
            raise NotImplementedError('row-sharded speaker passes support teacher feedback only (the point where every '
                                      'row has produced EOS, speaker.py:196, is a property of the whole batch)')

        # ---- encoder: Tp x (visual attention -> cat -> dropout -> LSTMCell), model.py:437-451
        ep = enc._params8()
        # (transposed copies of W_v / W_h, cached per weight version: the query q = W_v^T t_v becomes a K-contiguous
        # small product instead of the strided NN kernel: 6.4 instead of 15 us per path step)
        vw = _lib.VisualW(*(p_.data_ptr() for p_ in ep[0:4]), transposed(ep[2]).data_ptr(), transposed(ep[0]).data_ptr())
        lw = struct_of(_lib.LstmW, ep[4:8])
        st.e = dict(xin=new(Tp, B, 2 * F), alpha=new(Tp, B, V), t_v=new(Tp, B, D), q=new(Tp, B, F),
                    gates=new(Tp, B, 4 * H), hs=new(Tp + 1, B, H), cs=new(Tp + 1, B, H))
        st.words = torch.empty(S + 1, B, dtype=torch.int64, device=dev)
        st.ended = torch.empty(B, dtype=torch.uint8, device=dev)
        # the initial conditions of the pass in one launch: zero encoder state (model.py:420-427), <BOS> (speaker.py:137),
        # no row has ended (speaker.py:136)
        fill_regions((st.e['hs'][0], 0.0), (st.e['cs'][0], 0.0), (st.words[0], BOS), (st.ended, 0))
        d_enc = dropout_arg(*st.drop_enc)
        # the chosen-action embeddings of ALL path steps in one gather ([Tp*B] rows: the index arrays are
        # [Tp,B] contiguous), and -- without dropout -- one strided copy into the LSTM inputs of all steps
        # the chosen-action embeddings of ALL path steps in one launch (the index arrays are [Tp,B] contiguous): without
        # dropout straight into the first half of the LSTM inputs, with dropout into a dense copy the steps mask
        gat = lambda out, ld: call('sf_gather_path_actions', ptr(store.table), V, store.IMG, store.LOC, ptr(batch.vp),   # noqa: E731
                                   ptr(batch.act_view), ptr(batch.act_sincos), ptr(batch.act), Tp * B, ptr(out), ld, ws[2])
        if d_enc is None:
            st.e['act_emb'] = None
            gat(st.e['xin'], 2 * F)
        else:
            st.e['act_emb'] = new(Tp, B, F)
            gat(st.e['act_emb'], F)
        # eval mode: the LSTM cell also writes h_t into ctx[:, t] (its second output), no stacking copy afterwards
        st.ctx = new(B, Tp, H)
        # hidden states of all word steps stacked: hs_all[t] is step t's incoming h (h_init first), hs_all[t + 1] its
        # output -- the batched weight-gradient products read hs_all[0:S] as one [S*B, H] matrix
        st.hs_all = new(S + 1, B, H)
        st.h_init = st.hs_all[0]
        e2d = enc.encoder2decoder
        # all Tp path steps (attention -> [action | feature] -> dropout -> cell) and decoder_init = tanh(encoder2decoder(h))
        # (model.py:437-453) as ONE library call
        pano0 = store.pano(batch.vp, batch.view)
        enc_args = (byref(vw), byref(lw), ptr(e2d.weight), ptr(e2d.bias), byref(pano0), Tp, B, H, D,
                    ptr(st.e['xin']), ptr(st.e['alpha']), ptr(st.e['t_v']), ptr(st.e['q']), ptr(st.e['gates']), ptr(st.e['hs']),
                    ptr(st.e['cs']), ptr(st.ctx) if d_enc is None else None, ptr(st.e['act_emb']), ptr(st.h_init), d_enc,
                    st.site_rel) + tuple(ws)
        no_backward = not training and not (torch.is_grad_enabled() and any(
            p_.requires_grad for p_ in list(ep) + [e2d.weight, e2d.bias] + list(dec._params9())))
        if no_backward and self.fold_query:
            # inference: the attention query through the float64 fold M_v = W_v^T W_h (one product per path step instead
            # of two dependent ones, one rounding fewer); cached per weight version
            fold, st.e['fold'] = visual_query_fold(ep[0], ep[1], ep[2])
            call('sf_speaker_encoder_fwd_folded', byref(fold), *enc_args)
        else:
            call('sf_speaker_encoder_fwd', *enc_args)
        st.c_init = st.e['cs'][Tp]
        if d_enc is not None:
            ctx_raw = st.e['hs'][1:].permute(1, 0, 2).contiguous()      # [B,Tp,H]
            call('sf_dropout_copy', ptr(ctx_raw), Tp * H, B, Tp * H, ptr(st.ctx), Tp * H, d_enc,
                 2 * (st.site_rel + Tp) + 1, 0, ws[2])

        # ---- decoder: S x (embedding -> LSTMCell -> dropout -> attention -> vocab projection)
        shapes = dict(emb=(E,), gates=(4 * H,), c1=(H,), h1=(H,), cat2=(2 * H,), t_text=(H,),
                      alpha=(Tp,), h_tilde=(H,), logit=(ldv,))
        st.tape = {k: new(S, B, *shapes[k]) for k in _DEC_TAPE if k not in ('h1', 'c1')}
        st.tape['h1'] = st.hs_all[1:]
        st.cs_all = new(S + 1, B, H)              # cell states likewise: slot 0 = c_init, slot t + 1 = c1 of step t
        st.tape['c1'] = st.cs_all[1:]
        st.step_scores, st.nll_term, st.live = new(S, B), new(S, B), new(S, B)
        st.sum_cnt, st.gscale, st.loss_buf = new(S, 2), new(S), new(1)
        d_dec = dropout_arg(*st.drop_dec)
        st.targets = batch.instr_seq[:, :S].t().contiguous()              # [S,B] (speaker.py:163)
        params = list(ep) + [enc.encoder2decoder.weight, enc.encoder2decoder.bias] + list(dec._params9())
        differentiable = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        # a trainable (non-GloVe) embedding with a backward to follow: embedded (train mode: dropped, model.py:499-500)
        # words instead of rows of the cached input-product table
        st.dec_table = not (differentiable and trainable_embedding(dec))
        dw = dec._w_struct(table=st.dec_table)
        persistent = False
        st.teacher_path = False
        # (`persistent = False` rules out every persistent launch: what the fault fallback of run() relies on)
        if self.teacher_batched and self.persistent and st.feedback == 0 and st.dec_table:
            # teacher forcing: the recurrence alone in one persistent launch, the rest for all S*B rows at once
            st.cs_all[0].copy_(st.c_init)
            tp0 = _lib.SpkDecoderTape(*(st.tape[k].data_ptr() if (k != 'emb' or differentiable) else None
                                        for k in _DEC_TAPE))
            rc = _lib.lib.sf_speaker_teacher_fwd(
                byref(dw), B, E, H, Tp, vocab, S, PAD, EOS, ptr(st.targets), ptr(st.hs_all), ptr(st.cs_all), ptr(st.ctx),
                ptr(batch.path_mask), ptr(st.words), ptr(st.ended), ptr(st.step_scores), ptr(st.nll_term), ptr(st.live),
                byref(tp0), d_dec, st.site_rel, *ws)
            if rc != 2:                                   # SF_ERR_UNSUPPORTED: shapes outside the persistent recurrence
                _lib.check(rc, 'sf_speaker_teacher_fwd')
                st.teacher_path = True
        if not st.teacher_path and self.persistent and not training and not differentiable:
            # inference: all S word steps in one persistent launch (csrc/sf_persist.hip)
            rc = _lib.lib.sf_speaker_decode(
                byref(dw), B, H, Tp, vocab, S, st.feedback, PAD, EOS, ptr(st.targets), ptr(st.h_init),
                ptr(st.c_init), ptr(st.ctx), ptr(batch.path_mask), ptr(st.words), ptr(st.ended),
                ptr(st.step_scores), ptr(st.nll_term), ptr(st.live), ptr(st.tape['logit']),
                ptr(st.tape['alpha']), ptr(st.tape['h1']), ptr(st.tape['c1']),
                byref(smp) if st.feedback == 2 else None, *ws)
            if rc != 2:                                   # SF_ERR_UNSUPPORTED: shapes outside the kernel
                _lib.check(rc, 'sf_speaker_decode')
                persistent = True
        st.persistent = persistent           # (the whole word loop as one persistent launch: sf_speaker_decode)
        if not persistent and not st.teacher_path:
            st.cs_all[0].copy_(st.c_init)
            # the per-step word loop WITH its tape (training, or shapes outside the persistent kernel), one library call:
            # sf_speaker_decoder_fwd + sf_speaker_glue_fwd per word with no host work between the launches.  The embedded
            # words are only kept for the backward (dW_ih); the forward looks the input product up in the [vocab,4H]
            # table (sf_spk_decoder_w.xw_table)
            tp0 = _lib.SpkDecoderTape(*(st.tape[k].data_ptr() if (k != 'emb' or differentiable) else None
                                        for k in _DEC_TAPE))
            call('sf_speaker_words_fwd', byref(dw), B, E, H, Tp, vocab, S, st.feedback, PAD, EOS, ptr(st.targets),
                 ptr(st.h_init), ptr(st.c_init), ptr(st.ctx), ptr(batch.path_mask), ptr(st.words), ptr(st.ended),
                 ptr(st.step_scores), ptr(st.nll_term), ptr(st.live), byref(tp0), d_dec, st.site_rel,
                 byref(smp) if st.feedback == 2 else None, *ws)
        call('sf_reduce_terms', ptr(st.nll_term), ptr(st.live), S, B, ptr(st.sum_cnt), ws[2])
        if self.group is not None:
            torch.distributed.all_reduce(st.sum_cnt, group=self.group)
        # (the step means are added up to and including the first step at which every row has produced EOS,
        # speaker.py:192-197; gscale is 0 behind it)
        if self.group is not None:
            # a row shard sees only ITS rows' EOS: the step at which every row of the BATCH has ended (speaker.py:196) is
            # not a local property.  Teacher feedback (the only sharded mode): behind that step every target is PAD, the
            # reduced table's counts are 0 and its steps add nothing -- the plain per-step means are the batch's loss
            call('sf_loss_finalize', ptr(st.sum_cnt), S, ptr(st.loss_buf), ptr(st.gscale), ws[2])
        else:
            call('sf_speaker_loss_finalize', ptr(st.sum_cnt), ptr(st.words), EOS, S, B, ptr(st.loss_buf), ptr(st.gscale), ws[2])
        st.logits = st.tape['logit'][:, :, :vocab]
        st.h, st.c = st.tape['h1'][S - 1], st.tape['c1'][S - 1]
        if differentiable:
            st.loss = _SpeakerLossFn.apply(self, st, *params)
        else:
            st.loss = st.loss_buf.reshape(())
        return st

    def capture_training(self, batch, steps, optimizers=(), feedback='teacher'):
        """hipGraph of ONE WHOLE TRAINING ITERATION of the speaker (speaker.py:376-395: zero_grad, teacher-forced scoring
        pass with dropout, loss.backward(), optimizer steps).  Returns a runtime.TrainingGraph (see
        FollowerEngine.capture_training): the ~1 300 launches of an iteration cost 6.7 ms of host issue when issued one
        by one and nothing when replayed."""
        from .runtime import TrainingGraph
        if self.group is not None:
            raise NotImplementedError('capture_training: row-sharded passes are issued eagerly')
        opts = list(optimizers)

        def body():
            for o in opts:
                o.zero_grad()
            st = self.score(batch, steps, feedback, train=True)
            st.loss.backward()
            for o in opts:
                o.step()
            return st
        return TrainingGraph(self, body, opts, self.store.device)

    def run(self, batch, steps, feedback='teacher', train=None):
        """`score` + the fault check of the persistent word loop (include/sf_hip.h: sf_workspace_fault_offset): one
        host sync; if the launch gave up a bounded wait (co-residency lost to another process: its outputs are
        NaN-poisoned) the SAME pass -- same dropout / sampling sites -- is re-issued on the per-step kernels in this
        process.  Raises PersistentLaunchFault if a fault is still raised afterwards."""
        site = self.site_next
        st = self.score(batch, steps, feedback, train)
        dev = self.store.device
        bits = take_fault(dev)
        if bits:
            self.fallbacks += 1
            keep, self.persistent, self.site_next = self.persistent, False, site
            try:
                st = self.score(batch, steps, feedback, train)
            finally:
                self.persistent = keep
            again = take_fault(dev)
            if again:
                raise PersistentLaunchFault('fault bits %d, and %d after the per-step re-issue' % (bits, again))
        return st

    def _backward(self, st, dloss):
        enc, dec, store = self.encoder, self.decoder, self.store
        batch, S = st.batch, st.steps
        B, Tp, H, F, V, D, E, vocab, ldv = st.dims
        dev = store.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        ws = ws_args(dev)
        gscale = st.gscale * dloss.to(torch.float32)
        dw, dg = dec._w_struct(table=st.dec_table, bwd=True), dec._w_struct(grad=True, table=st.dec_table)
        d_dec = dropout_arg(*st.drop_dec)
        dlogit = new(B, ldv)
        dh_a, dc_a, dh_b, dc_b = new(B, H), new(B, H), new(B, H), new(B, H)
        dctx = torch.zeros(B, Tp, H, device=dev, dtype=torch.float32)
        # the backward of all S word steps (glue + decoder, last step first) as one library call
        tp0 = _lib.SpkDecoderTape(*(st.tape[k].data_ptr() for k in _DEC_TAPE))
        in_b = C.c_int(0)
        # per step only the data gradients; every weight gradient is ONE product over all S*B stacked rows at the end
        gt = dict(dlogit=new(S, B, ldv), dpre=new(S, B, H), dt_text=new(S, B, H), dgates=new(S, B, 4 * H))
        gtape = _lib.SpkDecoderGTape(*(gt[k].data_ptr() for k in ('dlogit', 'dpre', 'dt_text', 'dgates')))
        done = False
        if st.teacher_path and self.stacked_wgrad:
            # teacher-forced pass: head backward over all S*B rows + the recurrence's backward as one persistent launch
            scratch = (new(S, B, 2 * H), new(S, B, Tp), new(S, B, H))
            rc = _lib.lib.sf_speaker_teacher_bwd(
                byref(dw), byref(dg), B, E, H, Tp, vocab, S, PAD, ptr(st.words), ptr(st.targets), ptr(st.hs_all),
                ptr(st.cs_all), ptr(st.ctx), byref(tp0), ptr(gscale.contiguous()), ptr(dh_a), ptr(dc_a), ptr(dctx), d_dec,
                st.site_rel, byref(gtape), ptr(scratch[0]), ptr(scratch[1]), ptr(scratch[2]), *ws)
            if rc != 2:
                _lib.check(rc, 'sf_speaker_teacher_bwd')
                dh1, dc1, done = dh_a, dc_a, True
        if not done:
            call('sf_speaker_words_bwd', byref(dw), byref(dg), B, E, H, Tp, vocab, S, PAD, ptr(st.words), ptr(st.targets),
                 ptr(st.h_init), ptr(st.c_init), ptr(st.ctx), byref(tp0), ptr(gscale.contiguous()), ptr(dlogit), ptr(dh_a),
                 ptr(dc_a), ptr(dh_b), ptr(dc_b), ptr(dctx), byref(in_b), d_dec, st.site_rel,
                 byref(gtape) if self.stacked_wgrad else None, ptr(st.hs_all), *ws)
            dh1, dc1 = (dh_b, dc_b) if in_b.value else (dh_a, dc_a)
        # ---- encoder backward
        d_enc = dropout_arg(*st.drop_enc)
        ep = enc._params8()
        vw = _lib.VisualW(*(p_.data_ptr() for p_ in ep[0:4]), transposed(ep[2]).data_ptr(), transposed(ep[0]).data_ptr())
        vg = _lib.VisualW(*_grads(ep[0:4]))
        lw = _lib.LstmW(*(p_.data_ptr() for p_ in ep[4:8]), transposed(ep[4]).data_ptr(), transposed(ep[5]).data_ptr())
        lg = _lib.LstmW(*_grads(ep[4:8]))
        e2d = enc.encoder2decoder
        g_e2d = _grads((e2d.weight, e2d.bias))
        # through decoder_init = tanh(W h_Tp + b): dh1 here is d h_init
        dh = new(B, H)
        call('sf_linear_bwd', ptr(st.e['hs'][Tp]), H, ptr(e2d.weight), ptr(st.h_init), H, ptr(dh1), H,
             B, H, H, 1, ptr(dh), H, 0, g_e2d[0], g_e2d[1], *ws)
        dc_in = dc1                                         # c_init is the raw cell state (model.py:457)
        # through ctx = dropout(stack(h_1..h_Tp))
        dctx_raw = new(B, Tp, H)
        call('sf_dropout_copy', ptr(dctx), Tp * H, B, Tp * H, ptr(dctx_raw), Tp * H, d_enc,
             2 * (st.site_rel + Tp) + 1, 0, ws[2])
        dctx_t = dctx_raw.permute(1, 0, 2).contiguous()    # [Tp,B,H]
        dxin = new(B, 2 * F)
        dh_in, dh_out = dh, new(B, H)
        dc_bufs, k = (new(B, H), new(B, H)), 0
        for t in range(Tp - 1, -1, -1):
            # h_{t+1} also feeds ctx[:, t]
            call('sf_add_f32', ptr(dh_in), ptr(dctx_t[t]), B * H, ws[2])
            pano = store.pano(batch.vp[t], batch.view[t])
            call('sf_lstm_cell_bwd', byref(lw), byref(lg), B, 2 * F, H, ptr(st.e['xin'][t]), 2 * F,
                 ptr(st.e['hs'][t]), ptr(st.e['cs'][t]), ptr(st.e['cs'][t + 1]),
                 ptr(st.e['gates'][t]), ptr(dh_in), ptr(dc_in), ptr(dxin), 2 * F, ptr(dh_out),
                 ptr(dc_bufs[k]), *ws)
            dxin_f = C.c_void_p(dxin.data_ptr() + 4 * F)
            call('sf_visual_attention_bwd', byref(vw), byref(vg), byref(pano), B, H, D,
                 ptr(st.e['hs'][t]), ptr(st.e['alpha'][t]), ptr(st.e['t_v'][t]), dxin_f, 2 * F, d_enc,
                 2 * (st.site_rel + t), F, ptr(dh_out), *ws)
            dh_in, dh_out = dh_out, dh_in
            dc_in, k = dc_bufs[k], k ^ 1


# ------------------------------------------------------------------------------------------------
# Throughput path for configs[2] (data_augmentation_from_speaker.py:35-82: Seq2SeqSpeaker.test over 178 300 sampled
# trajectories, one minibatch after the other): packed index batches, double-buffered pinned staging, two streams.
# ------------------------------------------------------------------------------------------------
def packed_layout(B, Tp, Lmax):
    """Byte offsets of one index-form speaker minibatch inside ONE contiguous buffer (8-byte aligned sections):
    what the host packs, one H2D copy moves and DeviceSpeakerBatch.from_packed views on the device."""
    off, lay = 0, {}
    for name, n, isz in (('instr_seq', B * Lmax, 8), ('vp', Tp * B, 4), ('view', Tp * B, 4), ('act', Tp * B, 4),
                         ('act_view', Tp * B, 4), ('act_sincos', Tp * B * 4, 4), ('path_mask', B * Tp, 1)):
        lay[name] = (off, n * isz)
        off = (off + n * isz + 7) & ~7
    lay['bytes'] = off
    return lay


def pack_speaker_batch(sb, out, Lmax=80, Tp=None):
    """Writes a synth.SpeakerBatch-like index batch (instr, path_len, vp, view, act_view, act_heading, act_elevation,
    act_is_stop: speaker.py:68-121 in index form) into the uint8 numpy buffer `out` in `packed_layout` order.  Same
    values as DeviceSpeakerBatch.from_synth.  Returns (B, Tp)."""
    Tp_all, B = sb.vp.shape
    Tp = int(sb.path_len.max()) if Tp is None else Tp
    lay = packed_layout(B, Tp, Lmax)
    view = lambda name, dt, shape: out[lay[name][0]:lay[name][0] + lay[name][1]].view(dt).reshape(shape)   # noqa: E731
    seq = view('instr_seq', np.int64, (B, Lmax))
    seq[:] = PAD
    for i, inst in enumerate(sb.instr):                                   # follower.py:75-105, reverse=False
        n = min(len(inst), Lmax - 1)
        seq[i, :n] = inst[:n]
        if len(inst) + 1 <= Lmax:
            seq[i, n] = EOS
        else:
            seq[i, Lmax - 1] = inst[Lmax - 1]
    live = np.arange(Tp)[:, None] < sb.path_len[None, :]
    view('vp', np.int32, (Tp, B))[:] = np.where(live, sb.vp[:Tp], -1)
    view('view', np.int32, (Tp, B))[:] = sb.view[:Tp]
    view('act', np.int32, (Tp, B))[:] = live & ~sb.act_is_stop[:Tp]
    view('act_view', np.int32, (Tp, B))[:] = sb.act_view[:Tp]
    view('act_sincos', np.float32, (Tp, B, 4))[:] = cand_sincos(sb.act_heading[:Tp], sb.act_elevation[:Tp])
    view('path_mask', np.uint8, (B, Tp))[:] = (~live).T
    return B, Tp


def batch_from_packed(buf, B, Tp, Lmax=80, row0=0):
    """DeviceSpeakerBatch whose tensors are VIEWS of the device uint8 buffer `buf` (packed_layout order)."""
    lay = packed_layout(B, Tp, Lmax)
    view = lambda name, dt, shape: buf[lay[name][0]:lay[name][0] + lay[name][1]].view(dt).view(shape)   # noqa: E731
    return DeviceSpeakerBatch(instr_seq=view('instr_seq', torch.int64, (B, Lmax)),
                              path_mask=view('path_mask', torch.uint8, (B, Tp)),
                              vp=view('vp', torch.int32, (Tp, B)), view=view('view', torch.int32, (Tp, B)),
                              act=view('act', torch.int32, (Tp, B)), act_view=view('act_view', torch.int32, (Tp, B)),
                              act_sincos=view('act_sincos', torch.float32, (Tp, B, 4)), row0=row0)


class SpeakerSweep:
    """Greedy (or sampled) decoding of MANY path minibatches at full device rate.

    ONE stream by default since round 6: with two, the kernels of the two streams do not overlap -- the first kernel of a
    stream's graph waits for the other stream's persistent word loop (it holds a workgroup on every CU and the
    persistent launches of a process are serialised by a device-wide lock): 773 us average "duration" of
    gather_path_actions_kernel under rocprofv3 against 5.3 us on one stream (profiles/r06_e_speaker_sweep_streams.txt);
    what the second stream bought was copy / host overlap only (1.23 - 1.34 ms per minibatch against 1.34, box to box).
    `n_streams=2` remains available.

    Per stream and per path-step count Tp one captured hipGraph of the whole pass (encoder + the
    persistent word loop) over a static device staging buffer; per minibatch the host packs the index arrays into a
    pinned buffer (its own, one per stream and slot: packing minibatch n+1 overlaps the device work of n), ONE
    asynchronous H2D copy, a graph replay, and an asynchronous D2H copy of the words (int16) into a pinned result
    array.  The streams alternate, so the encoder kernels of one minibatch run beside the other's word loop -- a
    persistent launch of one 4-wave workgroup per CU that leaves most issue slots idle.  Every persistent launch is
    checked through the fault word at the end (runtime.take_fault); a sweep that saw a fault is re-run -- all of it -- on
    graphs captured with the per-step kernels (`fallbacks` counts them).  `sample` feedback needs vocab <= 1024 (the
    two-level draw of sf_sampling.h); the pinned staging buffers grow with the longest path met."""

    def __init__(self, encoder, decoder, store, batch_size, words, feedback='argmax', Lmax=80, n_streams=1, slots=2,
                 with_scores=False):
        """with_scores: every minibatch also returns its per-word scores [S,B] and its per-step (sum, count) table [S,2]
        (teacher-forced scoring sweeps: Seq2SeqSpeaker._issue_scores)."""
        self.with_scores = with_scores
        self.enc, self.dec, self.store = encoder, decoder, store
        self.B, self.S, self.Lmax, self.feedback = batch_size, words, Lmax, feedback
        self.streams = [torch.cuda.Stream(device=store.device) for _ in range(n_streams)]
        self.slots = slots
        self.graphs = {}                 # (stream index, Tp, persistent) -> (replay, state, device staging buffer)
        self.fallbacks = 0               # sweeps re-run on the per-step kernels after a persistent-launch fault
        if feedback == 'sample' and decoder.vocab_size > 1024:
            raise NotImplementedError('sample feedback draws with the two-level sampler of sf_sampling.h: vocab <= 1024 '
                                      '(%d here)' % decoder.vocab_size)
        cap = packed_layout(batch_size, 16, Lmax)['bytes']
        self.pinned = [[torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(slots)] for _ in self.streams]
        self.free_ev = [[None] * slots for _ in self.streams]
        self.host_pack_s = 0.0

    def _graph(self, si, Tp, persistent=True):
        key = (si, Tp, persistent)
        if key not in self.graphs:
            dev = self.store.device
            buf = torch.zeros(packed_layout(self.B, Tp, self.Lmax)['bytes'], dtype=torch.uint8, device=dev)
            batch = batch_from_packed(buf, self.B, Tp, self.Lmax)
            batch.vp.fill_(0)
            batch.instr_seq.fill_(EOS)
            # (the staging buffer was just initialised on the CURRENT stream: the sweep stream must see that, or its
            # warm-up pass gathers with whatever indices the recycled memory holds)
            self.streams[si].wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self.streams[si]):
                eng = SpeakerEngine(self.enc, self.dec, self.store)
                eng.persistent = persistent
                replay, st = eng.capture(batch, self.S, self.feedback)
                words16 = torch.empty(self.S, self.B, dtype=torch.int16, device=dev)
            torch.cuda.synchronize(dev)
            self.graphs[key] = (replay, st, buf, words16)
        return self.graphs[key]

    def run(self, batches, _persistent=True):
        """`batches`: sequence of synth.SpeakerBatch-like index batches of `batch_size` paths.  Returns an int16 array
        [n, S, B] of generated word ids (pinned host memory); with_scores: (words, scores [n,S,B] f32, sum_cnt [n,S,2])."""
        return self.finish(self.issue(batches, _persistent))

    def issue(self, batches, _persistent=True):
        """Everything of `run` that does not wait: packs, copies and replays every minibatch on its stream and returns
        the handle `finish` collects (the host is free in between: Seq2SeqSpeaker's pragmatic scoring builds its result
        dictionaries there)."""
        import time
        dev = self.store.device
        n = len(batches)
        out = torch.empty(n, self.S, self.B, dtype=torch.int16).pin_memory()
        sc = torch.empty(n, self.S, self.B, dtype=torch.float32).pin_memory() if self.with_scores else None
        cnt = torch.empty(n, self.S, 2, dtype=torch.float32).pin_memory() if self.with_scores else None
        self.host_pack_s = 0.0
        from .runtime import take_fault
        take_fault(dev)
        for s_ in self.streams:                                  # (the weights' derived layouts are refreshed on the caller's stream)
            s_.wait_stream(torch.cuda.current_stream(dev))
        # a path-step count met for the first time is captured for EVERY stream at once (a capture is tens of
        # milliseconds: better in front of the sweep than in the middle of it)
        for Tp in sorted({int(sb.path_len.max()) for sb in batches}):
            for si in range(len(self.streams)):
                self._graph(si, Tp, _persistent)
        for i, sb in enumerate(batches):
            si, slot = i % len(self.streams), (i // len(self.streams)) % self.slots
            pin = self.pinned[si][slot]
            if self.free_ev[si][slot] is not None:
                self.free_ev[si][slot].synchronize()            # its previous H2D copy has left the buffer
            t0 = time.perf_counter()
            need = packed_layout(len(sb.instr), int(sb.path_len.max()), self.Lmax)['bytes']
            if need > pin.numel():                               # a longer path than any before: grow this slot
                pin = self.pinned[si][slot] = torch.empty(need, dtype=torch.uint8).pin_memory()
            B, Tp = pack_speaker_batch(sb, pin.numpy(), self.Lmax)
            self.host_pack_s += time.perf_counter() - t0
            if B != self.B:
                raise ValueError('SpeakerSweep was built for minibatches of %d paths, got %d' % (self.B, B))
            replay, st, buf, words16 = self._graph(si, Tp, _persistent)
            with torch.cuda.stream(self.streams[si]):
                buf.copy_(pin[:buf.numel()], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self.free_ev[si][slot] = ev
                replay()
                words16.copy_(st.words[1:])                      # int64 -> int16 on the device (ids < 2^15)
                out[i].copy_(words16, non_blocking=True)
                if self.with_scores:
                    sc[i].copy_(st.step_scores, non_blocking=True)
                    cnt[i].copy_(st.sum_cnt, non_blocking=True)
        return dict(batches=batches, out=out, scores=sc, cnt=cnt, persistent=_persistent)

    def finish(self, handle, check_faults=True):
        """Waits for the sweep `issue` started; a sweep that saw a persistent-launch fault is re-run -- all of it -- on
        the per-step kernels (check_faults=False: the caller reads the fault words itself, runtime.take_fault)."""
        from .runtime import take_fault
        dev = self.store.device
        for s in self.streams:
            s.synchronize()
        bits = take_fault(dev) if check_faults else 0
        if bits:
            if not handle['persistent']:
                raise PersistentLaunchFault('fault bits %d raised by a sweep on the per-step kernels' % bits)
            # a starved persistent launch poisoned some minibatch: the whole sweep again on the per-step kernels
            self.fallbacks += 1
            return self.finish(self.issue(handle['batches'], _persistent=False))
        if self.with_scores:
            return handle['out'].numpy(), handle['scores'].numpy(), handle['cnt'].numpy()
        return handle['out'].numpy()
