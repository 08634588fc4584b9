"""Oracle (test infrastructure): the hot path as differentiable torch-CPU fp32
functions, so that autograd supplies the reference gradients for backward parity.
See oracle/__init__.py for the rules.  Own code built from stock torch ops; each
function cites the reference lines it restates (paths under /root/reference).

Weights are dicts of torch tensors keyed like the reference state_dicts.
"""
import torch
import torch.nn.functional as F


def to_torch(state, requires_grad=False, frozen=()):
    out = {}
    for k, v in state.items():
        t = torch.tensor(v, dtype=torch.float32)
        if requires_grad and k not in frozen:
            t.requires_grad_(True)
        out[k] = t
    return out


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """nn.LSTMCell (gate order i,f,g,o) as used at model.py:393, 434, 515."""
    gates = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, g, o = gates.chunk(4, dim=1)
    c1 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h1 = torch.sigmoid(o) * torch.tanh(c1)
    return h1, c1


def soft_dot_attention(h, context, mask, w_in, w_out):
    """model.py:122-143.  mask True = masked; filled on .data in the reference
    (:135), i.e. outside autograd -- masked_fill gives the same gradients."""
    target = F.linear(h, w_in)
    attn = torch.bmm(context, target.unsqueeze(2)).squeeze(2)
    if mask is not None:
        attn = attn.masked_fill(mask, float('-inf'))
    attn = torch.softmax(attn, dim=1)
    weighted = torch.bmm(attn.unsqueeze(1), context).squeeze(1)
    h_tilde = torch.tanh(F.linear(torch.cat((weighted, h), 1), w_out))
    return h_tilde, attn


def visual_soft_dot_attention(h, visual_context, w_h, b_h, w_v, b_v):
    """model.py:310-326."""
    target = F.linear(h, w_h, b_h)
    context = F.linear(visual_context, w_v, b_v)
    attn = torch.softmax(torch.bmm(context, target.unsqueeze(2)).squeeze(2), dim=1)
    weighted = torch.bmm(attn.unsqueeze(1), visual_context).squeeze(1)
    return weighted, attn


def eltwise_prod_scoring(h, all_u, w_h, b_h, w_a, b_a, w_out, b_out):
    """model.py:342-352."""
    target = F.linear(h, w_h, b_h).unsqueeze(1)
    context = F.linear(all_u, w_a, b_a)
    return F.linear(target * context, w_out, b_out).squeeze(2)


def encoder_lstm(enc, seq, lengths, drop_ctx=None):
    """model.py:81-104 with packed-sequence semantics written out (rows stop
    advancing at their own length; ctx zero beyond).  drop_ctx: optional
    multiplicative mask for the ctx dropout at :102 (GloVe path: no input dropout)."""
    emb = enc['embedding.weight'][seq]
    B = seq.shape[0]
    H = enc['lstm.weight_hh_l0'].shape[1]
    T = int(max(lengths))
    lens = torch.as_tensor(lengths)
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    outs = []
    for t in range(T):
        h1, c1 = lstm_cell(emb[:, t], h, c, enc['lstm.weight_ih_l0'], enc['lstm.weight_hh_l0'],
                           enc['lstm.bias_ih_l0'], enc['lstm.bias_hh_l0'])
        live = (t < lens).unsqueeze(1)
        h = torch.where(live, h1, h)
        c = torch.where(live, c1, c)
        outs.append(torch.where(live, h1, torch.zeros_like(h1)))
    ctx = torch.stack(outs, dim=1)
    decoder_init = torch.tanh(F.linear(h, enc['encoder2decoder.weight'],
                                       enc['encoder2decoder.bias']))
    if drop_ctx is not None:
        ctx = ctx * drop_ctx
    return ctx, decoder_init, c


def attn_decoder_step(dec, u_prev, all_u, visual_context, h0, c0, ctx, ctx_mask,
                      drop_in=None, drop_h=None):
    """model.py:377-397."""
    p = 'visual_attention_layer.'
    feature, alpha_v = visual_soft_dot_attention(
        h0, visual_context, dec[p + 'linear_in_h.weight'], dec[p + 'linear_in_h.bias'],
        dec[p + 'linear_in_v.weight'], dec[p + 'linear_in_v.bias'])
    concat = torch.cat((u_prev, feature), 1)
    if drop_in is not None:
        concat = concat * drop_in
    h1, c1 = lstm_cell(concat, h0, c0, dec['lstm.weight_ih'], dec['lstm.weight_hh'],
                       dec['lstm.bias_ih'], dec['lstm.bias_hh'])
    h1_drop = h1 if drop_h is None else h1 * drop_h
    h_tilde, alpha = soft_dot_attention(
        h1_drop, ctx, ctx_mask, dec['text_attention_layer.linear_in.weight'],
        dec['text_attention_layer.linear_out.weight'])
    q = 'decoder2action.'
    logit = eltwise_prod_scoring(
        h_tilde, all_u, dec[q + 'linear_in_h.weight'], dec[q + 'linear_in_h.bias'],
        dec[q + 'linear_in_a.weight'], dec[q + 'linear_in_a.bias'],
        dec[q + 'linear_out.weight'], dec[q + 'linear_out.bias'])
    return h1, c1, alpha, logit, alpha_v


def follower_rollout(enc, dec, seq, lengths, ctx_mask, steps, step_inputs, targets,
                     feedback, dims_feat, drop_masks=None):
    """follower.py:430-539 without the simulator (see oracle.np_model.follower_rollout).
    drop_masks(t) -> (drop_in[B,2F], drop_h[B,H]) or None; drop_masks('ctx') -> [B,T,H]."""
    drop_ctx = drop_masks('ctx') if drop_masks else None
    ctx, h, c = encoder_lstm(enc, seq, lengths, drop_ctx)
    B = seq.shape[0]
    u_prev = torch.zeros(B, dims_feat)
    loss = torch.zeros(())
    seq_scores = torch.zeros(B)
    ended = torch.zeros(B, dtype=torch.bool)
    logits, actions = [], []
    for t in range(steps):
        X, all_u, is_valid = (torch.as_tensor(a) for a in step_inputs(t))
        d_in, d_h = drop_masks(t) if drop_masks else (None, None)
        h, c, alpha, logit, alpha_v = attn_decoder_step(dec, u_prev, all_u, X, h, c, ctx,
                                                        ctx_mask, d_in, d_h)
        logit = logit.masked_fill(is_valid == 0, float('-inf'))            # :477
        target = torch.where(ended, torch.full_like(targets[t], -1), targets[t])
        if bool((target != -1).any()):
            loss = loss + F.cross_entropy(logit, target, ignore_index=-1)  # :481
        if feedback == 'teacher':
            a_t = target.clamp(min=0)                                      # :486
        elif feedback == 'argmax':
            a_t = logit.argmax(dim=1)                                      # :488
        else:
            raise ValueError(feedback)
        u_prev = all_u[torch.arange(B), a_t].detach()                      # :502
        seq_scores = seq_scores - F.cross_entropy(logit, a_t, reduction='none').detach()
        logits.append(logit)
        actions.append(a_t)
        ended = ended | (a_t == 0)
        if bool(ended.all()):
            break
    return dict(logits=logits, actions=torch.stack(actions), loss=loss,
                scores=seq_scores, h=h, c=c, ctx=ctx)


def speaker_encoder(enc, action_embs, world_feats, drop_masks=None):
    """model.py:437-457.  drop_masks(t) -> mask for the concat input (:433);
    drop_masks('ctx') -> mask for :456."""
    B = world_feats[0].shape[0]
    H = enc['lstm.weight_hh'].shape[1]
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    hs = []
    p = 'visual_attention_layer.'
    for t, (a_emb, X) in enumerate(zip(action_embs, world_feats)):
        feature, _ = visual_soft_dot_attention(
            h, X, enc[p + 'linear_in_h.weight'], enc[p + 'linear_in_h.bias'],
            enc[p + 'linear_in_v.weight'], enc[p + 'linear_in_v.bias'])
        concat = torch.cat((a_emb, feature), 1)
        if drop_masks:
            concat = concat * drop_masks(t)
        h, c = lstm_cell(concat, h, c, enc['lstm.weight_ih'], enc['lstm.weight_hh'],
                         enc['lstm.bias_ih'], enc['lstm.bias_hh'])
        hs.append(h)
    decoder_init = torch.tanh(F.linear(h, enc['encoder2decoder.weight'],
                                       enc['encoder2decoder.bias']))
    ctx = torch.stack(hs, dim=1)
    if drop_masks:
        ctx = ctx * drop_masks('ctx')
    return ctx, decoder_init, c


def speaker_decoder_step(dec, prev_word, h0, c0, ctx, ctx_mask, drop_h=None):
    """model.py:497-519, GloVe (no embedding dropout), non-att-feed branch."""
    emb = dec['embedding.weight'][prev_word]
    h1, c1 = lstm_cell(emb, h0, c0, dec['lstm.weight_ih'], dec['lstm.weight_hh'],
                       dec['lstm.bias_ih'], dec['lstm.bias_hh'])
    h1_drop = h1 if drop_h is None else h1 * drop_h
    h_tilde, alpha = soft_dot_attention(
        h1_drop, ctx, ctx_mask, dec['attention_layer.linear_in.weight'],
        dec['attention_layer.linear_out.weight'])
    logit = F.linear(h_tilde, dec['decoder2action.weight'], dec['decoder2action.bias'])
    return h1, c1, alpha, logit


def speaker_score(enc, dec, action_embs, world_feats, path_mask, instr_seq, steps,
                  feedback, pad_idx=0, bos_idx=3, eos_idx=2):
    """speaker.py:135-197."""
    ctx, h, c = speaker_encoder(enc, action_embs, world_feats)
    B = ctx.shape[0]
    w_t = torch.full((B,), bos_idx, dtype=torch.long)
    ended = torch.zeros(B, dtype=torch.bool)
    loss = torch.zeros(())
    seq_scores = torch.zeros(B)
    words, logits = [], []
    for t in range(steps):
        h, c, alpha, logit = speaker_decoder_step(dec, w_t, h, c, ctx, path_mask)
        target = instr_seq[:, t]
        if feedback == 'teacher':
            w_t = target
        elif feedback == 'argmax':
            w_t = logit.argmax(dim=1)
        else:
            raise ValueError(feedback)
        logp = F.log_softmax(logit, dim=1)
        seq_scores = seq_scores - F.nll_loss(logp, w_t, ignore_index=pad_idx,
                                             reduction='none').detach()
        if bool((target != pad_idx).any()):
            loss = loss + F.nll_loss(logp, target, ignore_index=pad_idx)
        logits.append(logit)
        words.append(w_t)
        ended = ended | (w_t == eos_idx)
        if bool(ended.all()):
            break
    return dict(logits=logits, words=torch.stack(words), loss=loss, scores=seq_scores,
                ctx=ctx, h=h, c=c)
