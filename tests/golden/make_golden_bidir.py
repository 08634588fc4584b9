#!/usr/bin/env python3
"""Golden vectors for the BIDIRECTIONAL instruction encoder (run in the BUILD container only).

train.py:197-199 builds EncoderLSTM(..., hidden_size // 2, ..., bidirectional=True) when asked to; model.py:61-66 and
92-94 then run nn.LSTM(bidirectional=True) over the packed instructions, return ctx = dropout([forward | reverse]),
decoder_init = tanh(encoder2decoder([h_reverse ; h_forward])) and c_t = [c_reverse ; c_forward].

  g12_encoder_bidir_eval     the module alone (frozen embedding), eval mode, B = 24 ragged instructions:
                             ctx, decoder_init, c_t
  g12_follower_bidir_train   the encoder with a TRAINABLE embedding (glove=None: the embedded tokens are dropped once,
                             ahead of both directions) + AttnDecoderLSTM, train mode with this repo's counter-based
                             masks (as make_golden_hard.py), B = 16, 6 teacher-forced steps: loss, first-step logits,
                             gradient norms + sampled entries of every parameter of both directions

    python tests/golden/make_golden_bidir.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from speaker_follower_amd import synth            # noqa: E402
from oracle import np_env, rng as orng             # noqa: E402
from make_golden import import_reference, load, t, grads_summary      # noqa: E402
from make_golden_hard import MaskedDrop, DROP_SEED, ENC_SEED_XOR       # noqa: E402

EMB_STREAM_XOR = 0x40000000


def main():
    torch.manual_seed(0)
    ref_model, _, _ = import_reference()
    dims = synth.FULL
    H, F, E = dims.hidden, dims.feat, dims.word
    loc_table = np_env.static_loc_embeddings()
    out = {}

    # ---- the module alone, eval mode
    enc_w = synth.bidirectional_encoder_weights(717, dims)
    enc = ref_model.EncoderLSTM(dims.vocab, E, H // 2, 0, 0.5, bidirectional=True, glove=enc_w['embedding.weight'])
    load(enc, enc_w)
    B = 24
    r = np.random.default_rng(31)
    lens = np.sort(r.integers(1, 41, size=B))[::-1].copy()
    lens[0] = 40
    lens[-1] = 1                                           # a one-token instruction: both directions see only it
    seq = np.zeros((B, 80), np.int64)
    for b in range(B):
        seq[b, :lens[b]] = r.integers(4, dims.vocab, size=lens[b])
    with torch.no_grad():
        ctx, h, c = enc(t(seq), [int(x) for x in lens])
    out['g12_encoder_bidir_eval'] = dict(seq=seq, lengths=lens.astype(np.int64), ctx=ctx.numpy(), decoder_init=h.numpy(),
                                         c_t=c.numpy(), weight_seed=np.int64(717))
    print('encoder alone: ctx %s, |h| %.4f' % (tuple(ctx.shape), float(h.abs().mean())))

    out['g12_follower_bidir_train'] = follower_case(ref_model, torch.float32)
    # the same step evaluated by the reference in float64: the anchor of the absolute logit bound (tests/tol.py)
    f64 = follower_case(ref_model, torch.float64)
    out['g12_follower_bidir_train']['logits_first_f64'] = f64['logits_first']
    out['g12_follower_bidir_train']['loss_f64'] = np.float64(f64['loss'])

    for name, arrays in out.items():
        path = os.path.join(HERE, name + '.npz')
        with tempfile.NamedTemporaryFile(dir=HERE, suffix='.npz', delete=False) as f:
            np.savez_compressed(f, **arrays)
        os.replace(f.name, path)
        print('%-34s %8.1f KB  %d arrays' % (name, os.path.getsize(path) / 1024, len(arrays)))


def follower_case(ref_model, dtype):
    """Follower training step through the bidirectional encoder with a trainable embedding."""
    dims = synth.FULL
    H, F, E = dims.hidden, dims.feat, dims.word
    loc_table = np_env.static_loc_embeddings()
    fl = lambda x: t(x).to(dtype)  # noqa: E731
    enc_w = synth.bidirectional_encoder_weights(818, dims)
    _, dec_w = synth.follower_weights_peaky(515, dims)
    enc = ref_model.EncoderLSTM(dims.vocab, E, H // 2, 0, 0.5, bidirectional=True, glove=None)
    dec = ref_model.AttnDecoderLSTM(F, H, 0.5, feature_size=F)
    load(enc, enc_w)
    load(dec, dec_w)
    enc.to(dtype).train()
    dec.to(dtype).train()
    enc.drop, dec.drop = MaskedDrop(), MaskedDrop()
    B, S, NVP = 16, 6, 64
    fb = synth.follower_batch(seed=29, batch=B, steps=S, n_viewpoints=NVP, min_len=5, max_len=20, stop_prob=0.05)
    table = synth.feature_table(4, NVP)
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, 80, reverse=True)
    T = max(lens)
    rows = np.arange(B)
    site0 = 0
    seed_enc = DROP_SEED ^ ENC_SEED_XOR
    enc.drop.queue = [fl(orng.dropout_mask(seed_enc, site0 ^ EMB_STREAM_XOR, rows, 80 * E, 0.5).reshape(B, 80, E)),
                      fl(orng.dropout_mask(seed_enc, site0, rows, T * H, 0.5).reshape(B, T, H))]
    for st in range(S):
        dec.drop.queue += [fl(orng.dropout_mask(DROP_SEED, 2 * (site0 + st), rows, 2 * F, 0.5)),
                           fl(orng.dropout_mask(DROP_SEED, 2 * (site0 + st) + 1, rows, H, 0.5))]
    torch.set_default_dtype(dtype)                    # (the reference's init_state builds default-dtype zeros)
    ctx, h, c = enc(t(seq), list(lens))
    u_prev = torch.zeros(B, F)
    ended = np.zeros(B, bool)
    loss = 0
    logits = []
    for st in range(S):
        X, U, is_valid = np_env.dense_follower_step(table, loc_table, fb, st)
        h, c, alpha, logit, alpha_v = dec(u_prev, fl(U), fl(X), h, c, ctx, t(mask).bool())
        logit = logit.masked_fill(t(is_valid) == 0, -float('inf'))
        target = np.where(ended, -1, fb.target[st])
        if (target >= 0).any():
            loss = loss + torch.nn.functional.cross_entropy(logit, t(target), ignore_index=-1)
        a_t = np.maximum(target, 0)
        u_prev = fl(U)[np.arange(B), a_t].detach()
        ended |= (a_t == 0)
        logits.append(logit.detach().numpy().copy())
    assert not enc.drop.queue and not dec.drop.queue
    loss.backward()
    torch.set_default_dtype(torch.float32)
    grng = np.random.default_rng(1313)
    res = dict(loss=float(loss.detach()) if dtype == torch.float64 else np.float32(float(loss.detach())), logits_first=logits[0], n_steps=np.int64(S), batch_seed=np.int64(29),
               enc_weight_seed=np.int64(818), dec_weight_seed=np.int64(515), table_seed=np.int64(4),
               dropout_seed=np.int64(DROP_SEED), site0=np.int64(site0))
    res.update({'enc/' + k: v for k, v in grads_summary(enc, grng).items()})
    res.update({'dec/' + k: v for k, v in grads_summary(dec, grng).items()})
    print('follower: loss %.5f, |d w_hh reverse| %.4e, |d embedding| %.4e'
          % (float(loss), float(enc.lstm.weight_hh_l0_reverse.grad.norm()), float(enc.embedding.weight.grad.norm())))
    return res


if __name__ == '__main__':
    main()
