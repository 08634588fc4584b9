"""Drop-in mirror of the reference's tasks/R2R/model.py module API on the HIP path.

Same class names, constructor signatures, forward signatures / return tuples and
state_dict key names as the reference (cited per class), so `train.py` /
`train_speaker.py`-style code and released checkpoints (`<path>_enc`, `<path>_dec`,
follower.py:1022-1035) work against it.  The arithmetic runs in libsf_hip.so; the
torch.nn sub-modules below are *parameter containers only* (they give the right
state_dict keys and default initialisation) -- their own forward is never called.

Autograd: each op is a torch.autograd.Function whose backward calls the matching
C-ABI backward.  Weight gradients are accumulated by the kernels straight into
`param.grad` (see runtime.grad_ptr); the Functions therefore return None for weight
inputs.  Gradients are NOT propagated through returned attention weights (alpha,
alpha_v): the reference never differentiates them.

Not supported (raises): bidirectional / multi-layer EncoderLSTM, non-GloVe
(trainable) embeddings in training mode, SpeakerDecoderLSTM(use_input_att_feed=True)
-- none is enabled by any reference script.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import call
from .runtime import (require_gpu, ptr, f32, stream, ws_args, dropout_arg, struct_of, grad_ptr,
                      pano_dense, cands_dense, transposed, xw_table, weight_key, _v)

byref = C.byref


def try_cuda(obj):
    """tasks/R2R/utils.py:195-204."""
    return obj.cuda() if torch.cuda.is_available() else obj


def _grads(params):
    """Pointer struct members for in-place weight-gradient accumulation."""
    return [_v(grad_ptr(p)) for p in params]


class _Holder:
    """An attribute bag (cache slot owner)."""


class _DropState:
    """Per-module dropout bookkeeping: a seed and a call counter, so every forward in
    training mode draws a fresh, reproducible mask (site id = counter)."""

    def __init__(self, salt=0):
        self.seed = None
        self.salt = salt
        self.counter = 0

    def next(self, module, p):
        if not module.training or p <= 0:
            return 0.0, 0, 0
        if self.seed is None:
            self.seed = (torch.initial_seed() + 0x9E3779B1 * self.salt) & 0xFFFFFFFF
        c = self.counter
        self.counter += 1
        return p, self.seed, c


# ------------------------------------------------------------------------------------------------
# nn.Linear with optional tanh (model.py:99, 453, 518)
# ------------------------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lin, act, x, w, b):
        y = ops.linear_fwd(x, w, b, act)
        ctx.lin, ctx.act = lin, act
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        lin = ctx.lin
        if lin.weight.requires_grad:
            grad_ptr(lin.weight)
        if lin.bias is not None and lin.bias.requires_grad:
            grad_ptr(lin.bias)
        dx = ops.linear_bwd(x, lin.weight.detach(), y, dy.contiguous(), ctx.act,
                            need_dx=ctx.needs_input_grad[2],
                            dw=lin.weight.grad if lin.weight.requires_grad else None,
                            db=lin.bias.grad if (lin.bias is not None and lin.bias.requires_grad)
                            else None)
        return None, None, dx, None, None


class _PaddedLinearFn(torch.autograd.Function):
    """nn.Linear whose output width is not a multiple of 4 (decoder2action: vocab 991): computed with the weight and bias
    padded by zero rows to the next multiple of 4 -- the kernels read rows of the output gradient 16 bytes at a time --
    and sliced; the padded gradients' first rows are added to the parameters' gradients."""

    @staticmethod
    def forward(ctx, lin, x, w, b):
        N, K = w.shape
        Np = (N + 3) & ~3
        wp = torch.zeros(Np, K, device=w.device, dtype=torch.float32)
        wp[:N].copy_(w.detach())
        bp = None
        if b is not None:
            bp = torch.zeros(Np, device=w.device, dtype=torch.float32)
            bp[:N].copy_(b.detach())
        y = ops.linear_fwd(x, wp, bp, 0)
        ctx.lin, ctx.N = lin, N
        ctx.save_for_backward(x, wp, y)
        return y[:, :N]

    @staticmethod
    def backward(ctx, dy):
        x, wp, y = ctx.saved_tensors
        lin, N = ctx.lin, ctx.N
        dyp = torch.zeros_like(y)
        dyp[:, :N].copy_(dy)
        dwp = torch.zeros_like(wp) if lin.weight.requires_grad else None
        dbp = torch.zeros(wp.shape[0], device=wp.device) if (lin.bias is not None and lin.bias.requires_grad) else None
        dx = ops.linear_bwd(x, wp, y, dyp, 0, need_dx=ctx.needs_input_grad[1], dw=dwp, db=dbp)
        if dwp is not None:
            grad_ptr(lin.weight)
            lin.weight.grad.add_(dwp[:N])
        if dbp is not None:
            grad_ptr(lin.bias)
            lin.bias.grad.add_(dbp[:N])
        return None, dx, None, None


def linear(lin, x, act=0):
    require_gpu(x)
    if lin.weight.shape[0] % 4 and act == 0:
        return _PaddedLinearFn.apply(lin, x.contiguous(), lin.weight, lin.bias)
    return _LinearFn.apply(lin, act, x.contiguous(), lin.weight, lin.bias)


class _DropoutFn(torch.autograd.Function):
    """Counter-based dropout over a [B, N] view (same mask forward and backward)."""

    @staticmethod
    def forward(ctx, x, p, seed, site):
        ctx.cfg = (p, seed, site)
        return _dropout_apply(x, p, seed, site)

    @staticmethod
    def backward(ctx, dy):
        p, seed, site = ctx.cfg
        return _dropout_apply(dy.contiguous(), p, seed, site), None, None, None


def _dropout_apply(x, p, seed, site):
    B = x.shape[0]
    x2 = x.reshape(B, -1)
    out = torch.empty_like(x2)
    call('sf_dropout_copy', ptr(x2), x2.shape[1], B, x2.shape[1], ptr(out), x2.shape[1],
         dropout_arg(p, seed), site, 0, stream())
    return out.view_as(x)


# ------------------------------------------------------------------------------------------------
class _SoftDotFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, h, context, mask, w_in, w_out):
        h_tilde, alpha, cat2, t_text = ops.soft_dot_attention_fwd((w_in, w_out), h, context, mask)
        ctx.mod = mod
        ctx.save_for_backward(context, alpha, cat2, t_text, h_tilde)
        ctx.mark_non_differentiable(alpha)
        return h_tilde, alpha

    @staticmethod
    def backward(ctx, dh_tilde, _dalpha):
        context, alpha, cat2, t_text, h_tilde = ctx.saved_tensors
        mod = ctx.mod
        ws = (mod.linear_in.weight, mod.linear_out.weight)
        for p in ws:
            grad_ptr(p)
        dh, dctx = ops.soft_dot_attention_bwd(
            tuple(p.detach() for p in ws), tuple(p.grad if p.requires_grad else None for p in ws),
            context, alpha, cat2, t_text, h_tilde, dh_tilde.contiguous(),
            need_dctx=ctx.needs_input_grad[2])
        return None, dh, dctx, None, None, None


class SoftDotAttention(nn.Module):
    """model.py:107-143.  forward(h, context, mask=None) -> (h_tilde, attn)."""

    def __init__(self, dim):
        super().__init__()
        self.linear_in = nn.Linear(dim, dim, bias=False)
        self.linear_out = nn.Linear(dim * 2, dim, bias=False)

    def forward(self, h, context, mask=None):
        require_gpu(h, context)
        return _SoftDotFn.apply(self, h.contiguous(), context.contiguous(), mask,
                                self.linear_in.weight, self.linear_out.weight)


class _TextAttnFn(torch.autograd.Function):
    """The attention core of ContextOnlySoftDotAttention (model.py:166-177): t = linear_in(h) -> (weighted context,
    attn)."""

    @staticmethod
    def forward(ctx, t, context, mask):
        B, L, H = context.shape
        alpha = torch.empty(B, L, device=t.device, dtype=torch.float32)
        wc = torch.empty(B, H, device=t.device, dtype=torch.float32)
        call('sf_text_attention_fwd', ptr(context), ptr(mask), B, L, H, ptr(t), H, ptr(alpha), ptr(wc), H, stream())
        ctx.save_for_backward(t, context, alpha)
        ctx.mark_non_differentiable(alpha)
        return wc, alpha

    @staticmethod
    def backward(ctx, dwc, _dalpha):
        t, context, alpha = ctx.saved_tensors
        B, L, H = context.shape
        dt = torch.empty_like(t)
        dctx = torch.zeros_like(context) if ctx.needs_input_grad[1] else None
        call('sf_text_attention_bwd', ptr(context), B, L, H, ptr(dwc.contiguous()), H, ptr(t), H, ptr(alpha), ptr(dt), H,
             ptr(dctx), stream())
        return dt, dctx, None


class ContextOnlySoftDotAttention(nn.Module):
    """model.py:146-177: like SoftDotAttention without the concatenation / tanh: forward(h, context, mask=None) ->
    (weighted_context, attn)."""

    def __init__(self, dim, context_dim=None):
        super().__init__()
        self.linear_in = nn.Linear(dim, dim if context_dim is None else context_dim, bias=False)

    def forward(self, h, context, mask=None):
        require_gpu(h, context)
        t = linear(self.linear_in, h)                                  # :166
        return _TextAttnFn.apply(t, context.contiguous(), ops.mask_u8(mask))


class _LstmCellFn(torch.autograd.Function):
    """nn.LSTMCell (model.py:505): (x, h0, c0) -> (h1, c1)."""

    @staticmethod
    def forward(ctx, cell, x, h0, c0, *params):
        h1, c1, gates = ops.lstm_cell_fwd(tuple(p.detach() for p in params), x, h0, c0)
        ctx.cell = cell
        ctx.save_for_backward(x, h0, c0, c1, gates)
        return h1, c1

    @staticmethod
    def backward(ctx, dh1, dc1):
        x, h0, c0, c1, gates = ctx.saved_tensors
        cell = ctx.cell
        ps = (cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
        for p in ps:
            grad_ptr(p)
        zero = lambda t, like: torch.zeros_like(like) if t is None else t.contiguous()   # noqa: E731
        dx, dh0, dc0 = ops.lstm_cell_bwd(tuple(p.detach() for p in ps), tuple(p.grad if p.requires_grad else None for p in ps),
                                         x, h0, c0, c1, gates, zero(dh1, h0), zero(dc1, c0), need_dx=True)
        return (None, dx, dh0, dc0) + (None,) * 4


def lstm_cell(cell, x, h0, c0):
    return _LstmCellFn.apply(cell, x.contiguous(), h0.contiguous(), c0.contiguous(), cell.weight_ih, cell.weight_hh,
                             cell.bias_ih, cell.bias_hh)


# ------------------------------------------------------------------------------------------------
class _VisualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, h, X, w_h, b_h, w_v, b_v):
        B, V, F = X.shape
        pano = pano_dense(X)
        out, alpha, t_v, q = ops.visual_attention_fwd((w_h, b_h, w_v, b_v), pano, B, V, F, h)
        ctx.mod = mod
        ctx.save_for_backward(h, X, alpha, t_v)
        ctx.mark_non_differentiable(alpha)
        return out, alpha

    @staticmethod
    def backward(ctx, dout, _dalpha):
        h, X, alpha, t_v = ctx.saved_tensors
        mod = ctx.mod
        ws = (mod.linear_in_h.weight, mod.linear_in_h.bias, mod.linear_in_v.weight,
              mod.linear_in_v.bias)
        for p in ws:
            grad_ptr(p)
        dh = ops.visual_attention_bwd(tuple(p.detach() for p in ws),
                                      tuple(p.grad if p.requires_grad else None for p in ws),
                                      pano_dense(X), X.shape[0], h, alpha, t_v, dout.contiguous())
        return None, dh, None, None, None, None, None


class VisualSoftDotAttention(nn.Module):
    """model.py:300-326.  forward(h, visual_context, mask=None) -> (weighted_context, attn)."""

    def __init__(self, h_dim, v_dim, dot_dim=256):
        super().__init__()
        self.linear_in_h = nn.Linear(h_dim, dot_dim, bias=True)
        self.linear_in_v = nn.Linear(v_dim, dot_dim, bias=True)

    def forward(self, h, visual_context, mask=None):
        require_gpu(h, visual_context)
        return _VisualFn.apply(self, h.contiguous(), f32(visual_context),
                               self.linear_in_h.weight, self.linear_in_h.bias,
                               self.linear_in_v.weight, self.linear_in_v.bias)


# ------------------------------------------------------------------------------------------------
class _ScoringFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, h, U, *w6):
        B, A, F = U.shape
        logit, t_a, wt, r = ops.eltwise_prod_scoring_fwd(w6, cands_dense(U), B, A, F, h)
        ctx.mod = mod
        ctx.save_for_backward(h, U, t_a, wt)
        return logit

    @staticmethod
    def backward(ctx, dlogit):
        h, U, t_a, wt = ctx.saved_tensors
        ws = ctx.mod._params6()
        for p in ws:
            grad_ptr(p)
        dh = ops.eltwise_prod_scoring_bwd(tuple(p.detach() for p in ws),
                                          tuple(p.grad if p.requires_grad else None for p in ws),
                                          cands_dense(U), U.shape[0], h, t_a, wt,
                                          dlogit.contiguous())
        return (None, dh, None) + (None,) * 6


class EltwiseProdScoring(nn.Module):
    """model.py:329-352.  forward(h, all_u_t, mask=None) -> logits [B, a_num]."""

    def __init__(self, h_dim, a_dim, dot_dim=256):
        super().__init__()
        self.linear_in_h = nn.Linear(h_dim, dot_dim, bias=True)
        self.linear_in_a = nn.Linear(a_dim, dot_dim, bias=True)
        self.linear_out = nn.Linear(dot_dim, 1, bias=True)

    def _params6(self):
        return (self.linear_in_h.weight, self.linear_in_h.bias, self.linear_in_a.weight,
                self.linear_in_a.bias, self.linear_out.weight, self.linear_out.bias)

    def forward(self, h, all_u_t, mask=None):
        require_gpu(h, all_u_t)
        return _ScoringFn.apply(self, h.contiguous(), f32(all_u_t), *self._params6())


# ------------------------------------------------------------------------------------------------
# EncoderLSTM (model.py:43-104)
# ------------------------------------------------------------------------------------------------
def trainable_embedding(mod):
    """True when a backward may follow that has to reach embedding.weight (glove=None, model.py:57-60): the input
    product is then formed from the embedded (and, in train mode, dropped: model.py:86-87) tokens instead of being
    read as a row of the cached [vocab,4H] table, and the embedded tokens are kept for the backward."""
    return mod.embedding.weight.requires_grad and torch.is_grad_enabled()


def _lstm_dir_params(mod, direction):
    sfx = '_reverse' if direction else ''
    return tuple(getattr(mod.lstm, n + '_l0' + sfx) for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'))


def _encoder_structs(mod, grad=False, seq=None, table=True, direction=None):
    """`table` = read the input product as rows of the cached [vocab,4H] table (False: form it from the embedded
    tokens, kept for the backward -- what a trainable embedding needs; callers decide with trainable_embedding() in
    THEIR grad mode: inside autograd.Function.forward / backward the grad mode says nothing).
    `direction` (0 forward / 1 reverse): one direction of a bidirectional encoder (SF_ENC_RAW_STATE: no encoder2decoder)."""
    l4 = _lstm_dir_params(mod, direction or 0)
    raw = direction is not None
    e2d = (None, None) if raw else (mod.encoder2decoder.weight, mod.encoder2decoder.bias)
    emb = mod.embedding
    if grad:
        ge = _grads((emb.weight,))[0] if (emb.weight.requires_grad and seq is not None) else None
        return _lib.EncoderG(_lib.LstmW(*_grads(l4)), *(_grads(e2d) if not raw else (None, None)), ge,
                             seq.data_ptr() if ge else None,
                             seq.shape[1] if ge else 0, emb.padding_idx if emb.padding_idx is not None else -1)
    owner = mod if not direction else mod._rev_cache         # (the cached table of the reverse direction lives apart)
    table = _xw_table(owner, emb.weight, l4[0]).data_ptr() if table else None
    flags = 0 if getattr(mod, 'persistent', True) else _lib.SF_ENC_PER_STEP
    if raw:
        flags |= _lib.SF_ENC_RAW_STATE | (_lib.SF_ENC_REVERSED if direction else 0)
    if table is None and not mod.use_glove:
        flags |= _lib.SF_ENC_EMB_DROPOUT               # (a no-op in eval mode: the dropout argument is NULL there)
    lw = _lib.LstmW(*(p.data_ptr() for p in l4), transposed(l4[0]).data_ptr() if table is None else None,
                    transposed(l4[1]).data_ptr())
    if raw:
        return _lib.EncoderW(emb.weight.data_ptr(), lw, None, None, None, table, flags)
    return _lib.EncoderW(emb.weight.data_ptr(), lw, *(p.data_ptr() for p in e2d),
                         transposed(e2d[0]).data_ptr(), table, flags)


def _xw_table(mod, emb, w_ih):
    """[vocab, 4H] = embedding W_ih^T (runtime.xw_table: cached on the module, rebuilt in place)."""
    return xw_table(mod, emb, w_ih)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, seq, lengths_dev, T, drop_cfg, use_table, direction, *params):
        B, Lpad = seq.shape
        E, H = mod.embedding_size, mod.hidden_size
        dev = seq.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        ctx_out, dinit, c_t = new(B, T, H), new(B, H), new(B, H)
        tape = dict(emb=new(T, B, E), xg=new(T, B, 4 * H), gates=new(T, B, 4 * H),
                    hs=new(T + 1, B, H), cs=new(T + 1, B, H))
        tp = _lib.EncoderTape(*(tape[k].data_ptr() for k in ('emb', 'xg', 'gates', 'hs', 'cs')))
        w = _encoder_structs(mod, table=use_table, direction=direction)
        p, seed, site = drop_cfg
        call('sf_encoder_lstm_fwd', byref(w), B, Lpad, T, E, H, ptr(seq), ptr(lengths_dev),
             ptr(ctx_out), ptr(dinit), ptr(c_t), byref(tp), dropout_arg(p, seed), site,
             *ws_args(dev))
        ctx.mod, ctx.tape, ctx.cfg = mod, tape, (B, T, E, H, drop_cfg, use_table, direction)
        ctx.n_params = len(params)
        ctx.save_for_backward(lengths_dev, dinit, seq)
        return ctx_out, dinit, c_t

    @staticmethod
    def backward(ctx, dctx, dinit_g, dct_g):
        lengths_dev, dinit, seq = ctx.saved_tensors
        mod, tape = ctx.mod, ctx.tape
        B, T, E, H, (p, seed, site), use_table, direction = ctx.cfg
        tp = _lib.EncoderTape(*(tape[k].data_ptr() for k in ('emb', 'xg', 'gates', 'hs', 'cs')))
        w = _encoder_structs(mod, table=use_table, direction=direction)
        g = _encoder_structs(mod, grad=True, seq=None if use_table else seq, direction=direction)
        cont = lambda t: t.contiguous() if t is not None else None  # noqa: E731
        dctx, dinit_g, dct_g = cont(dctx), cont(dinit_g), cont(dct_g)
        call('sf_encoder_lstm_bwd', byref(w), byref(g), B, T, E, H, ptr(lengths_dev), ptr(dinit),
             ptr(dctx), ptr(dinit_g), ptr(dct_g), byref(tp), dropout_arg(p, seed), site,
             *ws_args(dinit.device))
        return (None,) * (7 + ctx.n_params)


class _BiAssembleFn(torch.autograd.Function):
    """model.py:92-102 for nn.LSTM(bidirectional=True): ctx[b, t] = [forward output at t | reverse output at t] (the
    reverse direction ran over the row's tokens in reversed order, so its output for position t is its step
    len_b - 1 - t), zeros beyond the row's length; h_t = [h_reverse ; h_forward], c_t likewise.  Pure data movement
    through the C ABI (row-strided copies and an index gather); the backward is the transposed movement."""

    @staticmethod
    def forward(ctx, idx_rev, ctx_f, ctx_r, h_f, h_r, c_f, c_r):
        B, T, Hd = ctx_f.shape
        dev = ctx_f.device
        s = stream()
        out = torch.empty(B, T, 2 * Hd, device=dev)
        call('sf_dropout_copy', ptr(ctx_f), Hd, B * T, Hd, ptr(out), 2 * Hd, None, 0, 0, s)
        call('sf_gather_rows', ptr(ctx_r), Hd, ptr(idx_rev), B * T, Hd, C.c_void_p(out.data_ptr() + 4 * Hd), 2 * Hd, s)
        h_t, c_t = torch.empty(B, 2 * Hd, device=dev), torch.empty(B, 2 * Hd, device=dev)
        for dst, first, second in ((h_t, h_r, h_f), (c_t, c_r, c_f)):              # model.py:93-94: [-1] (reverse) first
            call('sf_dropout_copy', ptr(first), Hd, B, Hd, ptr(dst), 2 * Hd, None, 0, 0, s)
            call('sf_dropout_copy', ptr(second), Hd, B, Hd, C.c_void_p(dst.data_ptr() + 4 * Hd), 2 * Hd, None, 0, 0, s)
        ctx.save_for_backward(idx_rev)
        ctx.dims = (B, T, Hd)
        return out, h_t, c_t

    @staticmethod
    def backward(ctx, dout, dh_t, dc_t):
        (idx_rev,) = ctx.saved_tensors
        B, T, Hd = ctx.dims
        dev = idx_rev.device
        s = stream()
        dout, dh_t, dc_t = dout.contiguous(), dh_t.contiguous(), dc_t.contiguous()
        dctx_f, dctx_r = torch.empty(B, T, Hd, device=dev), torch.empty(B, T, Hd, device=dev)
        call('sf_dropout_copy', ptr(dout), 2 * Hd, B * T, Hd, ptr(dctx_f), Hd, None, 0, 0, s)
        # (the reversal is an involution on the live positions; idx < 0 -- beyond the row's length -- gives zeros)
        call('sf_gather_rows', C.c_void_p(dout.data_ptr() + 4 * Hd), 2 * Hd, ptr(idx_rev), B * T, Hd, ptr(dctx_r), Hd, s)
        outs = []
        for d in (dh_t, dc_t):
            first, second = torch.empty(B, Hd, device=dev), torch.empty(B, Hd, device=dev)
            call('sf_dropout_copy', ptr(d), 2 * Hd, B, Hd, ptr(first), Hd, None, 0, 0, s)
            call('sf_dropout_copy', C.c_void_p(d.data_ptr() + 4 * Hd), 2 * Hd, B, Hd, ptr(second), Hd, None, 0, 0, s)
            outs += [second, first]                                   # (forward half, reverse half)
        return None, dctx_f, dctx_r, outs[0], outs[1], outs[2], outs[3]


class EncoderLSTM(nn.Module):
    """model.py:43-104.  forward(inputs [B,L] int64, lengths) -> (ctx [B,maxlen,H],
    decoder_init [B,H], c_t [B,H])."""

    def __init__(self, vocab_size, embedding_size, hidden_size, padding_idx, dropout_ratio,
                 bidirectional=False, num_layers=1, glove=None):
        super().__init__()
        if num_layers != 1:
            raise NotImplementedError('HIP EncoderLSTM: a single layer only (the reference scripts never pass another '
                                      'num_layers, train.py:197-199)')
        self.embedding_size = embedding_size
        self.hidden_size = hidden_size
        self.drop = nn.Dropout(p=dropout_ratio)
        self.num_directions = 2 if bidirectional else 1
        self.num_layers = 1
        self.embedding = nn.Embedding(vocab_size, embedding_size, padding_idx)
        self.use_glove = glove is not None
        if self.use_glove:
            print('Using GloVe embedding')
            self.embedding.weight.data[...] = torch.from_numpy(glove)
            self.embedding.weight.requires_grad = False
        self.lstm = nn.LSTM(embedding_size, hidden_size, 1, batch_first=True, bidirectional=bidirectional)
        self.encoder2decoder = nn.Linear(hidden_size * self.num_directions, hidden_size * self.num_directions)
        self._drop_state = _DropState(1)
        self._rev_cache = _Holder()          # cached input-product table of the reverse direction (runtime.xw_table)

    def forward(self, inputs, lengths):
        require_gpu(inputs)
        lengths = [int(x) for x in lengths]
        T = max(lengths)
        lengths_dev = torch.tensor(lengths, dtype=torch.int32, device=inputs.device)
        cfg = self._drop_state.next(self, self.drop.p)
        table = not trainable_embedding(self)
        if self.num_directions == 2:
            return self._forward_bidirectional(inputs.contiguous(), lengths_dev, T, cfg, table)
        params = [self.lstm.weight_ih_l0, self.lstm.weight_hh_l0, self.lstm.bias_ih_l0,
                  self.lstm.bias_hh_l0, self.encoder2decoder.weight, self.encoder2decoder.bias,
                  self.embedding.weight]
        return _EncoderFn.apply(self, inputs.contiguous(), lengths_dev, T, cfg, table, None, *params)

    def _forward_bidirectional(self, seq, lengths_dev, T, cfg, table):
        """model.py:61-66, 88-102 with bidirectional=True (train.py:197-199: hidden_size // 2 per direction): two
        recurrences over the packed sequences -- the reverse one over every row's tokens in reversed order --, ctx =
        dropout([forward | reverse]), decoder_init = tanh(encoder2decoder([h_reverse ; h_forward])), c_t = [c_reverse ;
        c_forward].  Each direction is the unidirectional C entry (SF_ENC_RAW_STATE); per-step kernels (the persistent
        launch is built for hidden 512)."""
        B, Lpad = seq.shape
        dev = seq.device
        ln = lengths_dev.view(B, 1).long()
        t_ = torch.arange(Lpad, device=dev).view(1, Lpad)
        src = torch.where(t_ < ln, ln - 1 - t_, t_)                    # position read by step t of the reverse direction
        seq_rev = torch.gather(seq, 1, src).contiguous()
        tt = torch.arange(T, device=dev).view(1, T)
        base = torch.arange(B, device=dev).view(B, 1) * T
        idx_rev = torch.where(tt < ln, base + ln - 1 - tt, torch.full_like(tt, -1)).to(torch.int32).reshape(-1).contiguous()
        outs = []
        for direction, s_ in ((0, seq), (1, seq_rev)):                 # (cfg: only the embedded tokens' dropout here)
            params = list(_lstm_dir_params(self, direction)) + [self.embedding.weight]
            outs.append(_EncoderFn.apply(self, s_, lengths_dev, T, cfg, table, direction, *params))
        (ctx_f, h_f, c_f), (ctx_r, h_r, c_r) = outs
        ctx_raw, h_t, c_t = _BiAssembleFn.apply(idx_rev, ctx_f, ctx_r, h_f, h_r, c_f, c_r)
        decoder_init = linear(self.encoder2decoder, h_t, act=1)        # model.py:99
        p, seed, site = cfg
        ctx_out = _DropoutFn.apply(ctx_raw, p, seed, site) if p > 0 else ctx_raw     # model.py:101-102
        return ctx_out, decoder_init, c_t


# ------------------------------------------------------------------------------------------------
# AttnDecoderLSTM (model.py:355-397)
# ------------------------------------------------------------------------------------------------
_TAPE_KEYS = ('t_v', 'q', 'alpha_v', 'xin', 'gates', 'c1', 'h1', 'cat2', 't_text', 'alpha',
              'h_tilde', 't_a', 'wt', 'r', 'logit')


def decoder_tape(B, H, F, D, V, L, A, device):
    new = lambda *s: torch.empty(*s, device=device, dtype=torch.float32)  # noqa: E731
    shapes = dict(t_v=(B, D), q=(B, F), alpha_v=(B, V), xin=(B, 2 * F), gates=(B, 4 * H),
                  c1=(B, H), h1=(B, H), cat2=(B, 2 * H), t_text=(B, H), alpha=(B, L),
                  h_tilde=(B, H), t_a=(B, D), wt=(B, D), r=(B, F), logit=(B, A))
    return {k: new(*shapes[k]) for k in _TAPE_KEYS}


def tape_struct(tape):
    return _lib.DecoderTape(*(tape[k].data_ptr() for k in _TAPE_KEYS))


def decoder_params(mod):
    """The 16 decoder tensors in sf_decoder_w order."""
    v, t, a = mod.visual_attention_layer, mod.text_attention_layer, mod.decoder2action
    return (mod.lstm.weight_ih, mod.lstm.weight_hh, mod.lstm.bias_ih, mod.lstm.bias_hh,
            v.linear_in_h.weight, v.linear_in_h.bias, v.linear_in_v.weight, v.linear_in_v.bias,
            t.linear_in.weight, t.linear_out.weight,
            a.linear_in_h.weight, a.linear_in_h.bias, a.linear_in_a.weight, a.linear_in_a.bias,
            a.linear_out.weight, a.linear_out.bias)


def decoder_fold(mod):
    """sf_decoder_fold for inference (see include/sf_hip.h), rebuilt only when one of the nine
    tensors it is made of changed.  Kept on the module so that the pointers stay alive."""
    params = decoder_params(mod)
    src = [params[i] for i in (4, 5, 6, 10, 11, 12, 13, 14, 15)]
    key = weight_key(*src)
    cached = getattr(mod, '_sf_fold', None)
    if cached is not None and cached[0] == key:
        return cached[2]
    H, F, D = mod.hidden_size, mod.feature_size, params[4].shape[0]
    dev = params[0].device
    new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
    # rebuilt IN PLACE (a captured hipGraph keeps pointing at these buffers: FollowerEngine.capture)
    bufs = cached[1] if cached is not None and cached[1][0].device == dev else \
        (new(F, H), new(F), new(F + 4, H), new(F + 4))
    w = decoder_w_struct(params)
    call('sf_decoder_fold_build', byref(w), H, D, F, *(ptr(b) for b in bufs), *ws_args(dev))
    fold = _lib.DecoderFold(*(b.data_ptr() for b in bufs))
    mod._sf_fold = (key, bufs, fold)
    return fold


def decoder_w_struct(params, grad=False, fold=None):
    if grad:
        vals = _grads(params)
        return _lib.DecoderW(_lib.LstmW(*vals[0:4]), _lib.VisualW(*vals[4:8]),
                             _lib.SoftdotW(*vals[8:10]), _lib.ScoringW(*vals[10:16]))
    vals = [p.data_ptr() for p in params]
    t = lambda i: transposed(params[i]).data_ptr()  # noqa: E731  (cached per weight version)
    return _lib.DecoderW(_lib.LstmW(*vals[0:4], t(0), t(1)),
                         _lib.VisualW(*vals[4:8], t(6), t(4)),
                         _lib.SoftdotW(*vals[8:10], t(8), t(9)),
                         _lib.ScoringW(*vals[10:16], t(12), t(10)),
                         C.pointer(fold) if fold is not None else None)


class _DecoderStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, drop_cfg, u_prev, all_u, X, h0, c0, context, mask, *params):
        B, V, F = X.shape
        A = all_u.shape[1]
        H = h0.shape[1]
        L = context.shape[1]
        D = params[4].shape[0]
        tape = decoder_tape(B, H, F, D, V, L, A, X.device)
        tp = tape_struct(tape)
        w = decoder_w_struct(params)
        pano, cnd = pano_dense(X), cands_dense(all_u)
        p, seed, site = drop_cfg
        call('sf_attn_decoder_fwd', byref(w), byref(pano), byref(cnd), B, H, D, L, ptr(u_prev),
             ptr(h0), ptr(c0), ptr(context), ptr(mask), None, byref(tp), None, dropout_arg(p, seed),
             site,
             *ws_args(X.device))
        ctx.mod, ctx.tape, ctx.cfg = mod, tape, (B, H, D, L, drop_cfg)
        ctx.save_for_backward(all_u, X, h0, c0, context)
        ctx.mark_non_differentiable(tape['alpha'], tape['alpha_v'])
        return tape['h1'], tape['c1'], tape['alpha'], tape['logit'], tape['alpha_v']

    @staticmethod
    def backward(ctx, dh1, dc1, _da, dlogit, _dav):
        all_u, X, h0, c0, context = ctx.saved_tensors
        B, H, D, L, (p, seed, site) = ctx.cfg
        params = decoder_params(ctx.mod)
        w, g = decoder_w_struct(params), decoder_w_struct(params, grad=True)
        tp = tape_struct(ctx.tape)
        dev = X.device
        cont = lambda t: t.contiguous() if t is not None else None  # noqa: E731
        dh1, dc1, dlogit = cont(dh1), cont(dc1), cont(dlogit)
        if dlogit is None:
            dlogit = torch.zeros_like(ctx.tape['logit'])
        dh0, dc0 = torch.empty_like(h0), torch.empty_like(c0)
        dctx = torch.zeros_like(context) if ctx.needs_input_grad[7] else None
        pano, cnd = pano_dense(X), cands_dense(all_u)
        call('sf_attn_decoder_bwd', byref(w), byref(g), byref(pano), byref(cnd), B, H, D, L,
             ptr(h0), ptr(c0), ptr(context), byref(tp), None, ptr(dlogit), ptr(dh1), ptr(dc1),
             ptr(dh0), ptr(dc0), ptr(dctx), dropout_arg(p, seed), site, *ws_args(dev))
        return (None, None, None, None, None, dh0, dc0, dctx, None) + (None,) * 16


class AttnDecoderLSTM(nn.Module):
    """model.py:355-397.  forward(u_t_prev, all_u_t, visual_context, h_0, c_0, ctx,
    ctx_mask=None) -> (h_1, c_1, alpha, logit, alpha_v)."""

    def __init__(self, embedding_size, hidden_size, dropout_ratio, feature_size=2048 + 128,
                 image_attention_layers=None):
        super().__init__()
        if embedding_size != feature_size:
            raise NotImplementedError('HIP AttnDecoderLSTM expects embedding_size == feature_size '
                                      '(train.py:32,38 uses 2176 for both)')
        self.embedding_size = embedding_size
        self.feature_size = feature_size
        self.hidden_size = hidden_size
        self.u_begin = try_cuda(torch.zeros(embedding_size))           # model.py:368-369
        self.drop = nn.Dropout(p=dropout_ratio)
        self.lstm = nn.LSTMCell(embedding_size + feature_size, hidden_size)
        self.visual_attention_layer = VisualSoftDotAttention(hidden_size, feature_size)
        self.text_attention_layer = SoftDotAttention(hidden_size)
        self.decoder2action = EltwiseProdScoring(hidden_size, embedding_size)
        self._drop_state = _DropState(2)

    def forward(self, u_t_prev, all_u_t, visual_context, h_0, c_0, ctx, ctx_mask=None):
        require_gpu(u_t_prev, all_u_t, visual_context, h_0, c_0, ctx)
        cfg = self._drop_state.next(self, self.drop.p)
        return _DecoderStepFn.apply(
            self, cfg, f32(u_t_prev), f32(all_u_t), f32(visual_context), h_0.contiguous(),
            c_0.contiguous(), ctx.contiguous(), ops.mask_u8(ctx_mask), *decoder_params(self))


# ------------------------------------------------------------------------------------------------
# Speaker (model.py:405-519)
# ------------------------------------------------------------------------------------------------
class _AttnLstmStepFn(torch.autograd.Function):
    """SpeakerEncoderLSTM._forward_one_step (model.py:429-435): visual attention ->
    cat(action_embedding, feature) -> dropout -> LSTMCell."""

    @staticmethod
    def forward(ctx, mod, drop_cfg, act_emb, X, h0, c0, *params):
        B, V, F = X.shape
        H = h0.shape[1]
        v = mod.visual_attention_layer
        D = v.linear_in_h.weight.shape[0]
        dev = X.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        xin, alpha, t_v, q = new(B, 2 * F), new(B, V), new(B, D), new(B, F)
        h1, c1, gates = new(B, H), new(B, H), new(B, 4 * H)
        p, seed, site = drop_cfg
        d = dropout_arg(p, seed)
        # (the speaker's path encoder forms its attention query and scores in float64 -- csrc/sf_precise.hip -- on the module
        # path as on the engines'; shapes outside that kernel take the fp32 one)
        vw = _lib.VisualW(*(p_.data_ptr() for p_ in params[0:4]), transposed(params[2]).data_ptr(),
                          transposed(params[0]).data_ptr())
        lw = struct_of(_lib.LstmW, params[4:8])
        pano = pano_dense(X)
        xin_f = C.c_void_p(xin.data_ptr() + 4 * F)
        rc = _lib.lib.sf_visual_attention_fwd_f64(byref(vw), byref(pano), B, H, D, ptr(h0), xin_f, 2 * F,
                                                  ptr(alpha), ptr(t_v), ptr(q), d, 2 * site, F, *ws_args(dev))
        if rc == 2:                                  # SF_ERR_UNSUPPORTED
            call('sf_visual_attention_fwd', byref(vw), byref(pano), B, H, D, ptr(h0), xin_f, 2 * F,
                 ptr(alpha), ptr(t_v), ptr(q), d, 2 * site, F, *ws_args(dev))
        else:
            _lib.check(rc, 'sf_visual_attention_fwd_f64')
        call('sf_dropout_copy', ptr(act_emb), F, B, F, ptr(xin), 2 * F, d, 2 * site, 0, stream())
        call('sf_lstm_cell_fwd', byref(lw), B, 2 * F, H, ptr(xin), 2 * F, ptr(h0), ptr(c0), ptr(h1),
             ptr(c1), ptr(gates), None, 0, None, 0, *ws_args(dev))
        ctx.mod, ctx.cfg = mod, (B, V, F, H, D, drop_cfg)
        ctx.save_for_backward(X, h0, c0, xin, alpha, t_v, c1, gates)
        return h1, c1

    @staticmethod
    def backward(ctx, dh1, dc1):
        X, h0, c0, xin, alpha, t_v, c1, gates = ctx.saved_tensors
        B, V, F, H, D, (p, seed, site) = ctx.cfg
        mod = ctx.mod
        params = mod._params8()
        d = dropout_arg(p, seed)
        dev = X.device
        vw, vg = struct_of(_lib.VisualW, params[0:4]), _lib.VisualW(*_grads(params[0:4]))
        lw, lg = struct_of(_lib.LstmW, params[4:8]), _lib.LstmW(*_grads(params[4:8]))
        cont = lambda t: t.contiguous() if t is not None else None  # noqa: E731
        dh1, dc1 = cont(dh1), cont(dc1)
        dxin = torch.empty_like(xin)
        dh0, dc0 = torch.empty_like(h0), torch.empty_like(c0)
        call('sf_lstm_cell_bwd', byref(lw), byref(lg), B, 2 * F, H, ptr(xin), 2 * F, ptr(h0), ptr(c0),
             ptr(c1), ptr(gates), ptr(dh1), ptr(dc1), ptr(dxin), 2 * F, ptr(dh0), ptr(dc0),
             *ws_args(dev))
        pano = pano_dense(X)
        dxin_f = C.c_void_p(dxin.data_ptr() + 4 * F)
        call('sf_visual_attention_bwd', byref(vw), byref(vg), byref(pano), B, H, D, ptr(h0),
             ptr(alpha), ptr(t_v), dxin_f, 2 * F, d, 2 * site, F, ptr(dh0), *ws_args(dev))
        return (None, None, None, None, dh0, dc0) + (None,) * 8


class SpeakerEncoderLSTM(nn.Module):
    """model.py:405-457.  forward(list of [B,F] action embeddings, list of [B,V,F]
    panoramas) -> (ctx [B,Tp,H], decoder_init [B,H], c [B,H])."""

    def __init__(self, action_embedding_size, world_embedding_size, hidden_size, dropout_ratio,
                 bidirectional=False):
        super().__init__()
        assert not bidirectional, 'Bidirectional is not implemented yet'
        if action_embedding_size != world_embedding_size:
            raise NotImplementedError('HIP SpeakerEncoderLSTM expects equal action / world sizes')
        self.action_embedding_size = action_embedding_size
        self.word_embedding_size = world_embedding_size
        self.hidden_size = hidden_size
        self.drop = nn.Dropout(p=dropout_ratio)
        self.visual_attention_layer = VisualSoftDotAttention(hidden_size, world_embedding_size)
        self.lstm = nn.LSTMCell(action_embedding_size + world_embedding_size, hidden_size)
        self.encoder2decoder = nn.Linear(hidden_size, hidden_size)
        self._drop_state = _DropState(3)

    def _params8(self):
        v = self.visual_attention_layer
        return (v.linear_in_h.weight, v.linear_in_h.bias, v.linear_in_v.weight, v.linear_in_v.bias,
                self.lstm.weight_ih, self.lstm.weight_hh, self.lstm.bias_ih, self.lstm.bias_hh)

    def init_state(self, batch_size):
        dev = self.lstm.weight_ih.device
        return (torch.zeros(batch_size, self.hidden_size, device=dev),
                torch.zeros(batch_size, self.hidden_size, device=dev))

    def _forward_one_step(self, h_0, c_0, action_embedding, world_state_embedding):
        cfg = self._drop_state.next(self, self.drop.p)
        return _AttnLstmStepFn.apply(self, cfg, f32(action_embedding), f32(world_state_embedding),
                                     h_0.contiguous(), c_0.contiguous(), *self._params8())

    def forward(self, batched_action_embeddings, world_state_embeddings):
        assert isinstance(batched_action_embeddings, list)
        assert isinstance(world_state_embeddings, list)
        assert len(batched_action_embeddings) == len(world_state_embeddings)
        require_gpu(world_state_embeddings[0])
        batch_size = world_state_embeddings[0].shape[0]
        h, c = self.init_state(batch_size)
        h_list = []
        for action_embedding, world_state_embedding in zip(batched_action_embeddings,
                                                           world_state_embeddings):
            h, c = self._forward_one_step(h, c, action_embedding, world_state_embedding)
            h_list.append(h)
        decoder_init = linear(self.encoder2decoder, h, act=1)          # model.py:453
        ctx = torch.stack(h_list, dim=1)                               # model.py:455
        p, seed, site = self._drop_state.next(self, self.drop.p)
        if p > 0:
            ctx = _DropoutFn.apply(ctx, p, seed, 2 * site + 1)         # model.py:456
        return ctx, decoder_init, c


_SPK_TAPE = ('emb', 'gates', 'c1', 'h1', 'cat2', 't_text', 'alpha', 'h_tilde', 'logit')


class _SpeakerDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, drop_cfg, use_table, prev_word, h0, c0, context, mask, *params):
        B, H = h0.shape
        Tp = context.shape[1]
        E, vocab = mod.vocab_embedding_size, mod.vocab_size
        ldv = (vocab + 3) & ~3
        dev = h0.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
        tape = dict(emb=new(B, E), gates=new(B, 4 * H), c1=new(B, H), h1=new(B, H),
                    cat2=new(B, 2 * H), t_text=new(B, H), alpha=new(B, Tp), h_tilde=new(B, H),
                    logit=new(B, ldv))
        tp = _lib.SpkDecoderTape(*(tape[k].data_ptr() for k in _SPK_TAPE))
        w = mod._w_struct(table=use_table)
        p, seed, site = drop_cfg
        call('sf_speaker_decoder_fwd', byref(w), B, E, H, Tp, vocab, ptr(prev_word), ptr(h0),
             ptr(c0), ptr(context), ptr(mask), None, byref(tp), dropout_arg(p, seed), site,
             *ws_args(dev))
        ctx.mod, ctx.tape, ctx.cfg = mod, tape, (B, E, H, Tp, vocab, ldv, drop_cfg, use_table)
        ctx.words = prev_word
        ctx.save_for_backward(h0, c0, context)
        ctx.mark_non_differentiable(tape['alpha'])
        return tape['h1'], tape['c1'], tape['alpha'], tape['logit'][:, :vocab]

    @staticmethod
    def backward(ctx, dh1, dc1, _da, dlogit):
        h0, c0, context = ctx.saved_tensors
        B, E, H, Tp, vocab, ldv, (p, seed, site), use_table = ctx.cfg
        mod = ctx.mod
        dev = h0.device
        w, g = mod._w_struct(table=use_table, bwd=True), mod._w_struct(grad=True, table=use_table)
        tp = _lib.SpkDecoderTape(*(ctx.tape[k].data_ptr() for k in _SPK_TAPE))
        dl = torch.zeros(B, ldv, device=dev, dtype=torch.float32)
        if dlogit is not None:
            dl[:, :vocab] = dlogit
        cont = lambda t: t.contiguous() if t is not None else None  # noqa: E731
        dh1, dc1 = cont(dh1), cont(dc1)
        dh0, dc0 = torch.empty_like(h0), torch.empty_like(c0)
        dctx = torch.zeros_like(context) if ctx.needs_input_grad[6] else None
        call('sf_speaker_decoder_bwd', byref(w), byref(g), B, E, H, Tp, vocab, ptr(ctx.words), ptr(h0), ptr(c0),
             ptr(context), byref(tp), ptr(dl), ptr(dh1), ptr(dc1), ptr(dh0), ptr(dc0), ptr(dctx),
             dropout_arg(p, seed), site, *ws_args(dev))
        return (None, None, None, None, dh0, dc0, dctx, None) + (None,) * 9


class SpeakerDecoderLSTM(nn.Module):
    """model.py:460-519.  forward(previous_word [B,1] int64, h_0, c_0, ctx, ctx_mask=None)
    -> (h_1, c_1, alpha, logit [B,vocab])."""

    def __init__(self, vocab_size, vocab_embedding_size, hidden_size, dropout_ratio, glove=None,
                 use_input_att_feed=False):
        super().__init__()
        self.vocab_size = vocab_size
        self.vocab_embedding_size = vocab_embedding_size
        self.hidden_size = hidden_size
        self.embedding = nn.Embedding(vocab_size, vocab_embedding_size)
        self.use_glove = glove is not None
        if self.use_glove:
            print('Using GloVe embedding')
            self.embedding.weight.data[...] = torch.from_numpy(glove)
            self.embedding.weight.requires_grad = False
        self.drop = nn.Dropout(p=dropout_ratio)
        self.use_input_att_feed = bool(use_input_att_feed)
        if self.use_input_att_feed:
            # model.py:475-481 (round 5; no reference script passes the flag): attention over the path context FIRST,
            # its output fed into the LSTM input.  Module-level path: every step is a composition of C-ABI operators
            # (Linear, text attention, LSTMCell, dropout) under torch autograd; the engines' fused word loops do not cover
            # this variant and step through the module (SpeakerEngine.score).
            print('using input attention feed in SpeakerDecoderLSTM')
            self.lstm = nn.LSTMCell(vocab_embedding_size + hidden_size, hidden_size)
            self.attention_layer = ContextOnlySoftDotAttention(hidden_size)
            self.output_l1 = nn.Linear(hidden_size * 2, hidden_size)
        else:
            self.lstm = nn.LSTMCell(vocab_embedding_size, hidden_size)
            self.attention_layer = SoftDotAttention(hidden_size)
        self.decoder2action = nn.Linear(hidden_size, vocab_size)
        self._drop_state = _DropState(4)

    def _params9(self):
        a = self.attention_layer
        return (self.embedding.weight, self.lstm.weight_ih, self.lstm.weight_hh, self.lstm.bias_ih,
                self.lstm.bias_hh, a.linear_in.weight, a.linear_out.weight,
                self.decoder2action.weight, self.decoder2action.bias)

    def _w_out_t(self):
        """decoder2action^T as [H, ldv] (padding columns zero), rebuilt in place when the weight changes: the backward's
        d h~ = dlogit W_out reads it K-contiguous."""
        w = self.decoder2action.weight
        key = weight_key(w)
        if getattr(self, '_wot_key', None) != key:
            V, H = w.shape
            ldv = (V + 3) & ~3
            buf = getattr(self, '_wot', None)
            if buf is None or buf.shape != (H, ldv) or buf.device != w.device:
                buf = torch.zeros(H, ldv, device=w.device, dtype=torch.float32)
            buf[:, :V].copy_(w.detach().t())
            self._wot, self._wot_key = buf, key
        return self._wot

    def _w_struct(self, grad=False, table=True, bwd=False):
        """`table`: see _encoder_structs (False = trainable embedding with a backward to follow).  `bwd`: with the
        transposed copies the backward's data gradients read K-contiguous (no strided NN products)."""
        ps = self._params9()
        if grad:
            v = _grads(ps[1:])
            ge = _grads(ps[0:1])[0] if (ps[0].requires_grad and not table) else None
            return _lib.SpkDecoderG(_lib.LstmW(*v[0:4]), _lib.SoftdotW(*v[4:6]), v[6], v[7], ge)
        v = [p.data_ptr() for p in ps]
        table = self._xw_table().data_ptr() if table else None
        flags = _lib.SF_SPK_EMB_DROPOUT if (table is None and not self.use_glove) else 0
        return _lib.SpkDecoderW(v[0], _lib.LstmW(*v[1:5], transposed(ps[1]).data_ptr() if table is None else None,
                                                 transposed(ps[2]).data_ptr() if bwd else None),
                                _lib.SoftdotW(v[5], v[6], transposed(ps[5]).data_ptr(), transposed(ps[6]).data_ptr() if bwd else None),
                                v[7], v[8], table, flags, self._w_out_t().data_ptr() if bwd else None)

    def _xw_table(self):
        """[vocab, 4H] = embedding W_ih^T (runtime.xw_table): the LSTM's input product becomes a
        row lookup by the previous word."""
        return xw_table(self, self.embedding.weight, self.lstm.weight_ih)

    def _forward_att_feed(self, previous_word, h_0, c_0, ctx, ctx_mask):
        """model.py:497-513, the use_input_att_feed branch.  Dropout sites 4 s + k of this module's s-th training call."""
        p, seed, site = self._drop_state.next(self, self.drop.p)
        drop = (lambda x, k: _DropoutFn.apply(x, p, seed, 4 * site + k)) if p > 0 else (lambda x, k: x)
        words = previous_word.reshape(-1)
        B = words.shape[0]
        emb = torch.empty(B, self.vocab_embedding_size, device=h_0.device, dtype=torch.float32)
        E = self.vocab_embedding_size
        call('sf_gather_rows', ptr(self.embedding.weight.detach()), E, ptr(words.to(torch.int32).contiguous()), B, E,
             ptr(emb), E, stream())                                    # :497-498
        if not self.use_glove:
            if self.embedding.weight.requires_grad:
                raise NotImplementedError('use_input_att_feed with a trainable embedding')
            emb = drop(emb, 0)                                         # :499-500
        h_tilde, alpha = self.attention_layer(drop(h_0, 1), ctx, ctx_mask)        # :502-503
        concat_input = torch.cat((emb, drop(h_tilde, 2)), 1)           # :504
        h_1, c_1 = lstm_cell(self.lstm, concat_input, h_0, c_0)        # :505
        x = drop(torch.cat((h_1, h_tilde), 1), 3)                      # :506-507
        x = linear(self.output_l1, x, act=1)                           # :508-509
        logit = linear(self.decoder2action, x)                         # :510
        return h_1, c_1, alpha, logit

    def forward(self, previous_word, h_0, c_0, ctx, ctx_mask=None):
        require_gpu(previous_word, h_0, c_0, ctx)
        if self.use_input_att_feed:
            return self._forward_att_feed(previous_word, h_0, c_0, ctx, ctx_mask)
        cfg = self._drop_state.next(self, self.drop.p)
        words = previous_word.reshape(-1).contiguous()                 # model.py:497-498
        return _SpeakerDecoderFn.apply(self, cfg, not trainable_embedding(self), words, h_0.contiguous(), c_0.contiguous(),
                                       ctx.contiguous(), ops.mask_u8(ctx_mask), *self._params9())
