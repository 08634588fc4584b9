"""Oracle (test infrastructure): literal numpy fp32 restatement of the forward pass
of the reference's model modules and per-step agent glue.  See oracle/__init__.py.

Every function follows the cited reference lines operation by operation (same
operand order, no algebraic folding) so that it can serve as the checker for the
HIP kernels, which *do* reorder the arithmetic.  Weights are dicts keyed like the
reference state_dicts.  All arrays fp32, indices int64, masks bool (True = masked).
"""
import numpy as np

f32 = np.float32


def linear(x, w, b=None):
    """torch.nn.Linear: x @ w.T + b."""
    y = x @ w.T
    if b is not None:
        y = y + b
    return y.astype(f32, copy=False)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(f32)


def softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / np.sum(e, axis=axis, keepdims=True)).astype(f32)


def log_softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    s = x - m
    return (s - np.log(np.sum(np.exp(s), axis=axis, keepdims=True))).astype(f32)


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """nn.LSTMCell as used at model.py:393, 434, 515 (gate order i,f,g,o)."""
    H = h.shape[1]
    gates = linear(x, w_ih, b_ih) + linear(h, w_hh, b_hh)
    i = sigmoid(gates[:, 0:H])
    f = sigmoid(gates[:, H:2 * H])
    g = np.tanh(gates[:, 2 * H:3 * H])
    o = sigmoid(gates[:, 3 * H:4 * H])
    c1 = (f * c + i * g).astype(f32)
    h1 = (o * np.tanh(c1)).astype(f32)
    return h1, c1


def soft_dot_attention(h, context, mask, w_in, w_out):
    """SoftDotAttention.forward, model.py:122-143."""
    target = linear(h, w_in)                                  # :129
    attn = np.einsum('bld,bd->bl', context, target).astype(f32)   # :132
    if mask is not None:
        attn = np.where(mask, f32(-np.inf), attn)             # :135
    attn = softmax(attn, axis=1)                              # :136
    weighted = np.einsum('bl,bld->bd', attn, context).astype(f32)  # :139
    h_tilde = np.tanh(linear(np.concatenate((weighted, h), 1), w_out))  # :140-142
    return h_tilde.astype(f32), attn


def visual_soft_dot_attention(h, visual_context, w_h, b_h, w_v, b_v):
    """VisualSoftDotAttention.forward, model.py:310-326."""
    target = linear(h, w_h, b_h)                              # :316
    context = linear(visual_context, w_v, b_v)                # :317  [B,V,D]
    attn = np.einsum('bvd,bd->bv', context, target).astype(f32)    # :320
    attn = softmax(attn, axis=1)                              # :321
    weighted = np.einsum('bv,bvf->bf', attn, visual_context).astype(f32)  # :324
    return weighted, attn


def eltwise_prod_scoring(h, all_u, w_h, b_h, w_a, b_a, w_out, b_out):
    """EltwiseProdScoring.forward, model.py:342-352."""
    target = linear(h, w_h, b_h)[:, None, :]                  # :348
    context = linear(all_u, w_a, b_a)                         # :349
    eltprod = target * context                                # :350
    logits = linear(eltprod, w_out, b_out)[:, :, 0]           # :351
    return logits.astype(f32)


def encoder_lstm(enc, seq, lengths):
    """EncoderLSTM.forward (GloVe/eval mode: no dropout), model.py:81-104.

    seq [B,Lpad] int64 with PAD after each row's length; pack_padded_sequence
    semantics: row b is advanced for t < lengths[b] only; ctx is zero beyond."""
    emb = enc['embedding.weight'][seq]                        # :85
    B = seq.shape[0]
    H = enc['lstm.weight_hh_l0'].shape[1]
    T = int(max(lengths))
    lengths = np.asarray(lengths)
    h = np.zeros((B, H), f32)
    c = np.zeros((B, H), f32)
    ctx = np.zeros((B, T, H), f32)
    for t in range(T):                                        # :89-90
        h1, c1 = lstm_cell(emb[:, t], h, c, enc['lstm.weight_ih_l0'],
                           enc['lstm.weight_hh_l0'], enc['lstm.bias_ih_l0'],
                           enc['lstm.bias_hh_l0'])
        live = (t < lengths)[:, None]
        h = np.where(live, h1, h)
        c = np.where(live, c1, c)
        ctx[:, t] = np.where(live, h1, 0.0)                   # :101 pad_packed -> zeros
    decoder_init = np.tanh(linear(h, enc['encoder2decoder.weight'],
                                  enc['encoder2decoder.bias'])).astype(f32)   # :99
    return ctx, decoder_init, c                               # :104


def attn_decoder_step(dec, u_prev, all_u, visual_context, h0, c0, ctx, ctx_mask,
                      drop_in=None, drop_h=None):
    """AttnDecoderLSTM.forward, model.py:377-397.  drop_* are optional
    multiplicative dropout masks (already scaled by 1/(1-p)); None = eval mode."""
    p = 'visual_attention_layer.'
    feature, alpha_v = visual_soft_dot_attention(
        h0, visual_context, dec[p + 'linear_in_h.weight'], dec[p + 'linear_in_h.bias'],
        dec[p + 'linear_in_v.weight'], dec[p + 'linear_in_v.bias'])       # :389
    concat = np.concatenate((u_prev, feature), 1)                         # :391
    if drop_in is not None:
        concat = concat * drop_in                                          # :392
    h1, c1 = lstm_cell(concat, h0, c0, dec['lstm.weight_ih'], dec['lstm.weight_hh'],
                       dec['lstm.bias_ih'], dec['lstm.bias_hh'])           # :393
    h1_drop = h1 if drop_h is None else h1 * drop_h                        # :394
    h_tilde, alpha = soft_dot_attention(
        h1_drop, ctx, ctx_mask, dec['text_attention_layer.linear_in.weight'],
        dec['text_attention_layer.linear_out.weight'])                     # :395
    q = 'decoder2action.'
    logit = eltwise_prod_scoring(
        h_tilde, all_u, dec[q + 'linear_in_h.weight'], dec[q + 'linear_in_h.bias'],
        dec[q + 'linear_in_a.weight'], dec[q + 'linear_in_a.bias'],
        dec[q + 'linear_out.weight'], dec[q + 'linear_out.bias'])          # :396
    return h1, c1, alpha, logit, alpha_v


def cross_entropy_terms(logit, target, ignore_index):
    """Per-row -log_softmax(logit)[target]; 0 where target == ignore_index."""
    lsm = log_softmax(logit, axis=1)
    t = np.where(target == ignore_index, 0, target)
    nll = -lsm[np.arange(len(t)), t]
    return np.where(target == ignore_index, f32(0), nll).astype(f32)


def follower_glue(logit, is_valid, target, feedback):
    """follower.py:476-505: mask invalid candidates, CE(ignore_index=-1, mean over
    non-ignored rows), next action, per-sample score of the chosen action."""
    logit = np.where(is_valid == 0, f32(-np.inf), logit).astype(f32)      # :477
    terms = cross_entropy_terms(logit, target, -1)
    n_live = int(np.sum(target != -1))
    # :481.  The reference never sees n_live == 0 (it breaks out first, :533); synthetic
    # targets can, and the HIP glue defines that step's loss as 0 the same way.
    loss = f32(np.sum(terms, dtype=f32) / f32(n_live)) if n_live > 0 else f32(0)
    if feedback == 'teacher':
        a_t = np.maximum(target, 0)                                       # :486
    elif feedback == 'argmax':
        a_t = np.argmax(logit, axis=1).astype(np.int64)                   # :488 (first max)
    else:
        raise ValueError(feedback)
    scores = -cross_entropy_terms(logit, a_t, -1)                         # :504
    return logit, loss, a_t, scores


def follower_rollout(enc, dec, seq, lengths, ctx_mask, steps, step_inputs, targets,
                     feedback, dims_feat, early_exit=True):
    """Env-free restatement of Seq2SeqAgent._rollout_with_loss (follower.py:430-539)
    / _score_obs_actions_and_instructions (:342-428) over precomputed per-step
    observations.  step_inputs(t) -> (X[B,V,F], all_u[B,A,F], is_valid[B,A]).

    Returns dict(logits list, actions [T,B], loss, scores [B], h, c)."""
    ctx, h, c = encoder_lstm(enc, seq, lengths)                           # :446
    B = seq.shape[0]
    u_prev = np.zeros((B, dims_feat), f32)                                # :462 u_begin
    loss = f32(0)
    seq_scores = np.zeros(B, f32)
    ended = np.zeros(B, bool)
    logits, actions, losses = [], [], []
    for t in range(steps):
        X, all_u, is_valid = step_inputs(t)
        h, c, alpha, logit, alpha_v = attn_decoder_step(dec, u_prev, all_u, X, h, c,
                                                        ctx, ctx_mask)    # :473
        target = np.where(ended, -1, targets[t])                          # :322-328
        logit, l_t, a_t, sc = follower_glue(logit, is_valid, target, feedback)
        loss = f32(loss + l_t)
        losses.append(l_t)
        u_prev = all_u[np.arange(B), a_t]                                 # :502
        seq_scores += sc                                                  # :505
        logits.append(logit)
        actions.append(a_t)
        ended |= (a_t == 0)                                               # :527-530
        if early_exit and ended.all():                                    # :533
            break
    return dict(logits=logits, actions=np.stack(actions), loss=loss, losses=losses,
                scores=seq_scores, h=h, c=c, ctx=ctx)


def speaker_encoder(enc, action_embs, world_feats):
    """SpeakerEncoderLSTM.forward, model.py:437-457 (eval mode).  Lists over path
    steps of [B,F] action embeddings and [B,V,F] panoramas; padded steps are
    zeros and still advance the state (no length masking, model.py:445-451)."""
    B = world_feats[0].shape[0]
    H = enc['lstm.weight_hh'].shape[1]
    h = np.zeros((B, H), f32)
    c = np.zeros((B, H), f32)
    hs = []
    p = 'visual_attention_layer.'
    for a_emb, X in zip(action_embs, world_feats):
        feature, _ = visual_soft_dot_attention(
            h, X, enc[p + 'linear_in_h.weight'], enc[p + 'linear_in_h.bias'],
            enc[p + 'linear_in_v.weight'], enc[p + 'linear_in_v.bias'])    # :431
        concat = np.concatenate((a_emb, feature), 1)                       # :432
        h, c = lstm_cell(concat, h, c, enc['lstm.weight_ih'], enc['lstm.weight_hh'],
                         enc['lstm.bias_ih'], enc['lstm.bias_hh'])         # :434
        hs.append(h)
    decoder_init = np.tanh(linear(h, enc['encoder2decoder.weight'],
                                  enc['encoder2decoder.bias'])).astype(f32)    # :453
    ctx = np.stack(hs, axis=1)                                             # :455
    return ctx, decoder_init, c


def speaker_decoder_step(dec, prev_word, h0, c0, ctx, ctx_mask):
    """SpeakerDecoderLSTM.forward, non-att-feed branch, model.py:497-519 (eval)."""
    emb = dec['embedding.weight'][prev_word]                               # :497-498
    h1, c1 = lstm_cell(emb, h0, c0, dec['lstm.weight_ih'], dec['lstm.weight_hh'],
                       dec['lstm.bias_ih'], dec['lstm.bias_hh'])           # :515
    h_tilde, alpha = soft_dot_attention(
        h1, ctx, ctx_mask, dec['attention_layer.linear_in.weight'],
        dec['attention_layer.linear_out.weight'])                          # :517
    logit = linear(h_tilde, dec['decoder2action.weight'], dec['decoder2action.bias'])  # :518
    return h1, c1, alpha, logit


def context_only_soft_dot_attention(h, context, mask, w_in):
    """ContextOnlySoftDotAttention.forward, model.py:161-177."""
    target = linear(h, w_in)                                               # :166
    attn = np.einsum('bld,bd->bl', context, target).astype(f32)            # :169
    if mask is not None:
        attn = np.where(mask, f32(-np.inf), attn)                          # :172
    attn = softmax(attn, axis=1)                                           # :173
    weighted = np.einsum('bl,bld->bd', attn, context).astype(f32)          # :176
    return weighted, attn


def speaker_decoder_step_att_feed(dec, prev_word, h0, c0, ctx, ctx_mask):
    """SpeakerDecoderLSTM.forward, the use_input_att_feed branch, model.py:497-513 (eval mode: dropout is identity)."""
    emb = dec['embedding.weight'][prev_word]                               # :497-498
    h_tilde, alpha = context_only_soft_dot_attention(h0, ctx, ctx_mask, dec['attention_layer.linear_in.weight'])   # :502-503
    concat_input = np.concatenate((emb, h_tilde), 1)                       # :504
    h1, c1 = lstm_cell(concat_input, h0, c0, dec['lstm.weight_ih'], dec['lstm.weight_hh'],
                       dec['lstm.bias_ih'], dec['lstm.bias_hh'])           # :505
    x = np.concatenate((h1, h_tilde), 1)                                   # :506
    x = np.tanh(linear(x, dec['output_l1.weight'], dec['output_l1.bias'])).astype(f32)        # :508-509
    logit = linear(x, dec['decoder2action.weight'], dec['decoder2action.bias'])               # :510
    return h1, c1, alpha, logit


def speaker_score(enc, dec, action_embs, world_feats, path_mask, instr_seq, steps,
                  feedback, pad_idx=0, bos_idx=3, eos_idx=2):
    """Seq2SeqSpeaker._score_obs_actions_and_instructions, speaker.py:135-197."""
    ctx, h, c = speaker_encoder(enc, action_embs, world_feats)             # :135
    B = ctx.shape[0]
    w_t = np.full(B, bos_idx, np.int64)                                    # :137
    ended = np.zeros(B, bool)
    loss = f32(0)
    seq_scores = np.zeros(B, f32)
    words, logits = [], []
    for t in range(steps):
        h, c, alpha, logit = speaker_decoder_step(dec, w_t, h, c, ctx, path_mask)  # :159
        target = instr_seq[:, t]                                           # :163
        if feedback == 'teacher':
            w_t = target                                                   # :167
        elif feedback == 'argmax':
            w_t = np.argmax(logit, axis=1).astype(np.int64)                # :169
        else:
            raise ValueError(feedback)
        word_scores = -cross_entropy_terms(logit, w_t, pad_idx)            # :179-180
        seq_scores += word_scores                                          # :181
        terms = cross_entropy_terms(logit, target, pad_idx)
        n_live = int(np.sum(target != pad_idx))
        if n_live > 0:                                                     # :182
            loss = f32(loss + np.sum(terms, dtype=f32) / f32(n_live))
        logits.append(logit)
        words.append(w_t)
        ended |= (w_t == eos_idx)                                          # :190-191
        if ended.all():                                                    # :196
            break
    return dict(logits=logits, words=np.stack(words), loss=loss, scores=seq_scores,
                ctx=ctx, h=h, c=c)
