"""Functional (non-autograd) wrappers over the C ABI: allocate outputs with torch,
pass raw pointers.  One wrapper per entry point of include/sf_hip.h; the modules
in model.py and the rollout engines build on these.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import call
from .runtime import (require_gpu, ptr, f32, stream, ws_args, dropout_arg, struct_of, _v,
                      pano_dense, cands_dense)

byref = C.byref


def _opt(s):
    return byref(s) if s is not None else None


# ------------------------------------------------------------------------------- nn.Linear
def linear_fwd(x, w, b=None, act=0):
    require_gpu(x, w)
    x, w = f32(x), f32(w)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    call('sf_linear_fwd', ptr(x), K, ptr(w), ptr(b), M, N, K, int(act), ptr(y), N,
         *ws_args(x.device))
    return y


def linear_bwd(x, w, y, dy, act=0, need_dx=True, dw=None, db=None):
    """dw/db: tensors accumulated in place (or None)."""
    x, w, dy = f32(x), f32(w), f32(dy)
    M, K = x.shape
    N = w.shape[0]
    dx = torch.empty_like(x) if need_dx else None
    call('sf_linear_bwd', ptr(x), K, ptr(w), ptr(y), N, ptr(dy), N, M, N, K, int(act), ptr(dx), K,
         0, ptr(dw), ptr(db), *ws_args(x.device))
    return dx


# ------------------------------------------------------------------------------- nn.LSTMCell
def lstm_cell_fwd(w4, x, h0, c0):
    require_gpu(x, h0, c0)
    x, h0, c0 = f32(x), f32(h0), f32(c0)
    B, I = x.shape
    H = h0.shape[1]
    h1 = torch.empty_like(h0)
    c1 = torch.empty_like(c0)
    gates = torch.empty(B, 4 * H, device=x.device, dtype=torch.float32)
    ws = struct_of(_lib.LstmW, w4)
    call('sf_lstm_cell_fwd', byref(ws), B, I, H, ptr(x), I, ptr(h0), ptr(c0), ptr(h1), ptr(c1),
         ptr(gates), None, 0, None, 0, *ws_args(x.device))
    return h1, c1, gates


def lstm_cell_bwd(w4, g4, x, h0, c0, c1, gates, dh1, dc1, need_dx=True):
    B, I = x.shape
    H = h0.shape[1]
    dx = torch.empty_like(x) if need_dx else None
    dh0 = torch.empty_like(h0)
    dc0 = torch.empty_like(c0)
    ws = struct_of(_lib.LstmW, w4)
    gs = struct_of(_lib.LstmW, g4) if g4 is not None else None
    call('sf_lstm_cell_bwd', byref(ws), _opt(gs), B, I, H, ptr(x), I, ptr(h0), ptr(c0), ptr(c1),
         ptr(gates), ptr(dh1), ptr(dc1), ptr(dx), I, ptr(dh0), ptr(dc0), *ws_args(x.device))
    return dx, dh0, dc0


# ------------------------------------------------------------------------------- VisualSoftDotAttention
def visual_attention_fwd(w4, pano, B, V, F, h, drop=None, drop_stream=0):
    """w4 = (w_h, b_h, w_v, b_v); pano: _lib.Pano.  Returns out [B,F], alpha [B,V], t_v, q."""
    require_gpu(h)
    h = f32(h)
    H = h.shape[1]
    D = w4[0].shape[0]
    dev = h.device
    out = torch.empty(B, F, device=dev, dtype=torch.float32)
    alpha = torch.empty(B, V, device=dev, dtype=torch.float32)
    t_v = torch.empty(B, D, device=dev, dtype=torch.float32)
    q = torch.empty(B, F, device=dev, dtype=torch.float32)
    ws = struct_of(_lib.VisualW, w4)
    call('sf_visual_attention_fwd', byref(ws), byref(pano), B, H, D, ptr(h), ptr(out), F,
         ptr(alpha), ptr(t_v), ptr(q), drop, drop_stream, 0, *ws_args(dev))
    return out, alpha, t_v, q


def visual_attention_bwd(w4, g4, pano, B, h, alpha, t_v, dout, drop=None, drop_stream=0):
    h, dout = f32(h), f32(dout)
    H = h.shape[1]
    D = w4[0].shape[0]
    dh = torch.zeros_like(h)
    ws = struct_of(_lib.VisualW, w4)
    gs = struct_of(_lib.VisualW, g4) if g4 is not None else None
    call('sf_visual_attention_bwd', byref(ws), _opt(gs), byref(pano), B, H, D, ptr(h), ptr(alpha),
         ptr(t_v), ptr(dout), dout.shape[1], drop, drop_stream, 0, ptr(dh), *ws_args(h.device))
    return dh


# ------------------------------------------------------------------------------- SoftDotAttention
def soft_dot_attention_fwd(w2, h, ctx, mask):
    require_gpu(h, ctx)
    h, ctx = f32(h), f32(ctx)
    B, L, H = ctx.shape
    dev = h.device
    h_tilde = torch.empty(B, H, device=dev, dtype=torch.float32)
    alpha = torch.empty(B, L, device=dev, dtype=torch.float32)
    cat2 = torch.empty(B, 2 * H, device=dev, dtype=torch.float32)
    t_text = torch.empty(B, H, device=dev, dtype=torch.float32)
    m = mask_u8(mask)
    ws = struct_of(_lib.SoftdotW, w2)
    call('sf_soft_dot_attention_fwd', byref(ws), B, L, H, ptr(h), H, ptr(ctx), ptr(m), None,
         ptr(h_tilde),
         ptr(alpha), ptr(cat2), ptr(t_text), *ws_args(dev))
    return h_tilde, alpha, cat2, t_text


def soft_dot_attention_bwd(w2, g2, ctx, alpha, cat2, t_text, h_tilde, dh_tilde, need_dctx=True):
    B, L, H = ctx.shape
    dh = torch.empty(B, H, device=ctx.device, dtype=torch.float32)
    dctx = torch.zeros_like(ctx) if need_dctx else None
    ws = struct_of(_lib.SoftdotW, w2)
    gs = struct_of(_lib.SoftdotW, g2) if g2 is not None else None
    call('sf_soft_dot_attention_bwd', byref(ws), _opt(gs), B, L, H, ptr(ctx), ptr(alpha), ptr(cat2),
         ptr(t_text), ptr(h_tilde), ptr(f32(dh_tilde)), ptr(dh), H, ptr(dctx), *ws_args(ctx.device))
    return dh, dctx


# ------------------------------------------------------------------------------- EltwiseProdScoring
def eltwise_prod_scoring_fwd(w6, cands, B, A, F, h):
    require_gpu(h)
    h = f32(h)
    H = h.shape[1]
    D = w6[0].shape[0]
    dev = h.device
    logit = torch.empty(B, A, device=dev, dtype=torch.float32)
    t_a = torch.empty(B, D, device=dev, dtype=torch.float32)
    wt = torch.empty(B, D, device=dev, dtype=torch.float32)
    r = torch.empty(B, F, device=dev, dtype=torch.float32)
    ws = struct_of(_lib.ScoringW, w6)
    call('sf_eltwise_prod_scoring_fwd', byref(ws), byref(cands), B, H, D, ptr(h), ptr(logit),
         ptr(t_a), ptr(wt), ptr(r), *ws_args(dev))
    return logit, t_a, wt, r


def eltwise_prod_scoring_bwd(w6, g6, cands, B, h, t_a, wt, dlogit):
    H = h.shape[1]
    D = w6[0].shape[0]
    dh = torch.empty_like(h)
    ws = struct_of(_lib.ScoringW, w6)
    gs = struct_of(_lib.ScoringW, g6) if g6 is not None else None
    call('sf_eltwise_prod_scoring_bwd', byref(ws), _opt(gs), byref(cands), B, H, D, ptr(h), ptr(t_a),
         ptr(wt), ptr(f32(dlogit)), ptr(dh), *ws_args(h.device))
    return dh


# ------------------------------------------------------------------------------- helpers
def mask_u8(mask):
    """bool / uint8 [B,L] mask -> contiguous uint8 (1 = masked), None passes through."""
    if mask is None:
        return None
    if mask.dtype == torch.bool:
        mask = mask.to(torch.uint8)
    elif mask.dtype != torch.uint8:
        raise TypeError('ctx_mask must be bool or uint8')
    return mask.contiguous()


def fill_(t, v=0.0):
    call('sf_fill_f32', ptr(t), t.numel(), float(v), stream())
    return t
