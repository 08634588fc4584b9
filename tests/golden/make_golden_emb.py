#!/usr/bin/env python3
"""Golden vectors for TRAINABLE word embeddings (run in the BUILD container only).

The reference modules built with glove=None (model.py:57-60, 470-473): the embedding is a parameter and the module's
dropout is also applied to the embedded tokens (model.py:86-87, 499-500).  As in make_golden_hard.py the reference's
nn.Dropout modules are replaced by one that applies the masks this repo's counter-based generator produces, so loss
and gradients -- the embedding's included -- are reproducible:

  g11_follower_trainable_emb   EncoderLSTM(glove=None) + AttnDecoderLSTM, train mode, B = 16, 6 teacher-forced steps:
                               loss, first-step logits, gradient norms + sampled entries of every parameter
  g11_speaker_trainable_emb    SpeakerDecoderLSTM(glove=None) alone, train mode, B = 12, 5 word steps on a random
                               context: per-step logits, loss = sum of NLL of the targets, gradients

    python tests/golden/make_golden_emb.py
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

EMB_STREAM_XOR = 0x40000000            # sf_hip.h: SF_ENC_EMB_DROPOUT site = ctx site ^ this


def main():
    torch.manual_seed(0)
    ref_model, _, _ = import_reference()
    dims = synth.FULL
    H, F, E = dims.hidden, dims.feat, dims.word
    loc_table = np_env.static_loc_embeddings()
    out = {}

    # ---- follower: encoder with a trainable embedding
    enc_w, dec_w = synth.follower_weights_peaky(515, dims)
    enc = ref_model.EncoderLSTM(dims.vocab, dims.word, dims.hidden, 0, 0.5, glove=None)
    dec = ref_model.AttnDecoderLSTM(dims.feat, dims.hidden, 0.5, feature_size=dims.feat)
    load(enc, enc_w)
    load(dec, dec_w)
    assert enc.embedding.weight.requires_grad
    enc.train()
    dec.train()
    enc.drop, dec.drop = MaskedDrop(), MaskedDrop()
    B, S, NVP = 16, 6, 64
    fb = synth.follower_batch(seed=23, batch=B, steps=S, n_viewpoints=NVP, min_len=5, max_len=20, stop_prob=0.05)
    table = synth.feature_table(4, NVP)
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, 80, reverse=True)
    T = max(lens)
    rows = np.arange(B)
    site0 = 0
    seed_enc = DROP_SEED ^ ENC_SEED_XOR
    enc.drop.queue = [t(orng.dropout_mask(seed_enc, site0 ^ EMB_STREAM_XOR, rows, 80 * E, 0.5).reshape(B, 80, E)),
                      t(orng.dropout_mask(seed_enc, site0, rows, T * H, 0.5).reshape(B, T, H))]
    for st in range(S):
        dec.drop.queue += [t(orng.dropout_mask(DROP_SEED, 2 * (site0 + st), rows, 2 * F, 0.5)),
                           t(orng.dropout_mask(DROP_SEED, 2 * (site0 + st) + 1, rows, H, 0.5))]
    ctx, h, c = enc(t(seq), list(lens))
    u_prev = torch.zeros(B, F)
    ended = np.zeros(B, bool)
    loss = 0
    logits = []
    for st in range(S):
        X, U, is_valid = np_env.dense_follower_step(table, loc_table, fb, st)
        h, c, alpha, logit, alpha_v = dec(u_prev, t(U), t(X), h, c, ctx, t(mask).bool())
        logit = logit.masked_fill(t(is_valid) == 0, -float('inf'))
        target = np.where(ended, -1, fb.target[st])
        if (target >= 0).any():
            loss = loss + torch.nn.functional.cross_entropy(logit, t(target), ignore_index=-1)
        a_t = np.maximum(target, 0)                                   # teacher feedback
        u_prev = t(U)[np.arange(B), a_t].detach()
        ended |= (a_t == 0)
        logits.append(logit.detach().numpy().copy())
    assert not enc.drop.queue and not dec.drop.queue
    loss.backward()
    grng = np.random.default_rng(1111)
    res = dict(loss=np.float32(float(loss)), logits_first=logits[0], n_steps=np.int64(S),
               batch_seed=np.int64(23), weight_seed=np.int64(515), table_seed=np.int64(4), dropout_seed=np.int64(DROP_SEED),
               site0=np.int64(site0))
    res.update({'enc/' + k: v for k, v in grads_summary(enc, grng).items()})
    res.update({'dec/' + k: v for k, v in grads_summary(dec, grng).items()})
    g_emb = enc.embedding.weight.grad.numpy()
    res['emb_grad_rows_nonzero'] = np.int64((np.abs(g_emb).sum(1) > 0).sum())
    res['emb_grad_row0_abs'] = np.float64(np.abs(g_emb[0]).sum())      # padding_idx row: exactly zero
    print('follower: loss %.5f, |d embedding| %.4e over %d token rows' % (float(loss), np.linalg.norm(g_emb), res['emb_grad_rows_nonzero']))
    out['g11_follower_trainable_emb'] = res

    # ---- speaker decoder with a trainable embedding, module level
    _, sdec_w = synth.speaker_weights_peaky(616, dims)
    sdec = ref_model.SpeakerDecoderLSTM(dims.vocab, dims.word, dims.hidden, 0.5, glove=None)
    load(sdec, sdec_w)
    sdec.train()
    sdec.drop = MaskedDrop()
    Bs, Ss, Tp = 12, 5, 6
    r = np.random.default_rng(77)
    ctx_s = (r.standard_normal((Bs, Tp, H)) * 0.5).astype(np.float32)
    pmask = np.zeros((Bs, Tp), bool)
    pmask[::3, 4:] = True
    h0 = (r.standard_normal((Bs, H)) * 0.3).astype(np.float32)
    c0 = (r.standard_normal((Bs, H)) * 0.3).astype(np.float32)
    words = r.integers(4, dims.vocab, size=(Ss + 1, Bs)).astype(np.int64)
    words[0] = 3
    seed_s, rows_s = 0xABCD, np.arange(Bs)
    for st in range(Ss):
        sdec.drop.queue += [t(orng.dropout_mask(seed_s, 2 * st, rows_s, E, 0.5)),
                            t(orng.dropout_mask(seed_s, 2 * st + 1, rows_s, H, 0.5))]
    h, c = t(h0), t(c0)
    loss = 0
    slog = []
    for st in range(Ss):
        h, c, alpha, logit = sdec(t(words[st]).view(-1, 1), h, c, t(ctx_s), t(pmask))
        loss = loss + torch.nn.functional.cross_entropy(logit, t(words[st + 1]))
        slog.append(logit.detach().numpy().copy())
    assert not sdec.drop.queue
    loss.backward()
    res = dict(loss=np.float32(float(loss)), logits=np.stack(slog), ctx=ctx_s, path_mask=pmask, h0=h0, c0=c0, words=words,
               weight_seed=np.int64(616), dropout_seed=np.int64(seed_s))
    res.update({'dec/' + k: v for k, v in grads_summary(sdec, np.random.default_rng(1212)).items()})
    print('speaker decoder: loss %.5f, |d embedding| %.4e' % (float(loss), float(sdec.embedding.weight.grad.norm())))
    out['g11_speaker_trainable_emb'] = res

    for name, arrays in out.items():
        path = os.path.join(HERE, name + '.npz')
        with tempfile.NamedTemporaryFile(dir=HERE, suffix='.npz', delete=False) as f:
            np.savez_compressed(f, **arrays)
        os.replace(f.name, path)
        print('%-34s %8.1f KB  %d arrays' % (name, os.path.getsize(path) / 1024, len(arrays)))


if __name__ == '__main__':
    main()
