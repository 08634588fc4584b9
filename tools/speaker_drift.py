#!/usr/bin/env python3
"""Where does the speaker's B = 100 logit distance from exact arithmetic come from?  (GPU; numpy float64 checker.)

Runs the G9 speaker case through the HIP path (per-step word loop and persistent word loop, split and fp32 gate
products) and compares, per word step,
  * trajectory distance: HIP h1 / c1 / logit against a float64 evaluation of the whole pass, and
  * LOCAL error: HIP's step t against the float64 evaluation of step t started from HIP's OWN h_{t-1}, c_{t-1}, ctx
    (what one step's arithmetic adds, nothing amplified).
    python tools/speaker_drift.py [teacher|argmax]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_follower_amd import synth, model, features, speaker, _lib   # noqa: E402
from oracle import np_env                                                  # noqa: E402  (checker)

f8 = np.float64


def sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def dec_step(W, w_prev, h, c, ctx, path_mask):
    H = h.shape[1]
    emb = W['embedding.weight'].astype(f8)[w_prev]
    gates = emb @ W['lstm.weight_ih'].astype(f8).T + W['lstm.bias_ih'] + h @ W['lstm.weight_hh'].astype(f8).T + W['lstm.bias_hh']
    i, f, g, o = sig(gates[:, :H]), sig(gates[:, H:2 * H]), np.tanh(gates[:, 2 * H:3 * H]), sig(gates[:, 3 * H:])
    c1 = f * c + i * g
    h1 = o * np.tanh(c1)
    tq = h1 @ W['attention_layer.linear_in.weight'].astype(f8).T
    att = np.einsum('bld,bd->bl', ctx, tq)
    att = np.where(path_mask, -np.inf, att)
    att = att - att.max(1, keepdims=True)
    e = np.exp(att)
    a = e / e.sum(1, keepdims=True)
    wc = np.einsum('bl,bld->bd', a, ctx)
    ht = np.tanh(np.concatenate((wc, h1), 1) @ W['attention_layer.linear_out.weight'].astype(f8).T)
    logit = ht @ W['decoder2action.weight'].astype(f8).T + W['decoder2action.bias']
    return dict(gates=gates, c1=c1, h1=h1, h_tilde=ht, logit=logit, alpha=a)


def encoder64(W, acts, feats):
    B, H = feats[0].shape[0], 512
    h, c = np.zeros((B, H)), np.zeros((B, H))
    hs = []
    p = 'visual_attention_layer.'
    for a_emb, X in zip(acts, feats):
        X = X.astype(f8)
        t = h @ W[p + 'linear_in_h.weight'].astype(f8).T + W[p + 'linear_in_h.bias']
        q = t @ W[p + 'linear_in_v.weight'].astype(f8)
        att = np.einsum('bvf,bf->bv', X, q)
        att = att - att.max(1, keepdims=True)
        e = np.exp(att)
        a = e / e.sum(1, keepdims=True)
        feat = np.einsum('bv,bvf->bf', a, X)
        x = np.concatenate((a_emb.astype(f8), feat), 1)
        gates = x @ W['lstm.weight_ih'].astype(f8).T + W['lstm.bias_ih'] + h @ W['lstm.weight_hh'].astype(f8).T + W['lstm.bias_hh']
        i, f, g, o = sig(gates[:, :H]), sig(gates[:, H:2 * H]), np.tanh(gates[:, 2 * H:3 * H]), sig(gates[:, 3 * H:])
        c = f * c + i * g
        h = o * np.tanh(c)
        hs.append(h)
    h0 = np.tanh(h @ W['encoder2decoder.weight'].astype(f8).T + W['encoder2decoder.bias'])
    return np.stack(hs, 1), h0, c


def main():
    feedback = sys.argv[1] if len(sys.argv) > 1 else 'teacher'
    g = np.load(os.path.join(ROOT, 'tests/golden/g9_speaker_b100_%s.npz' % feedback))
    d = synth.FULL
    senc_w, sdec_w = synth.speaker_weights_peaky(int(g['weight_seed']))
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    sb = synth.speaker_batch(seed=int(g['batch_seed']), batch=100, n_viewpoints=256, min_len=10, max_len=79)
    table = synth.feature_table(int(g['table_seed']), 256)
    store = features.FeatureStore(table)
    batch = speaker.DeviceSpeakerBatch.from_synth(sb)
    n = int(g['n_steps'])
    words = g['words']
    acts, feats, path_mask = np_env.dense_speaker_inputs(sb, table, np_env.static_loc_embeddings())
    ctx64, h064, c064 = encoder64(senc_w, acts, feats)
    # float64 trajectory on the golden's word sequence
    traj = []
    h, c = h064, c064
    w = np.full(100, 3, np.int64)
    for t in range(n):
        o = dec_step(sdec_w, w, h, c, ctx64, path_mask)
        traj.append(o)
        h, c = o['h1'], o['c1']
        w = words[t]
    modes = [('per-step words, split gates, float64 attention query', False, 0, 1),
             ('persistent words, split gates, float64 attention query', True, 0, 1),
             ('per-step words, split gates, fp32 attention (rounds 1-4)', False, 0, 0),
             ('persistent words, fp32 gates, fp32 attention', True, 1, 0)]
    for name, persistent, f32gate, precise in modes:
        _lib.lib.sf_debug_gate_product_f32(f32gate)
        _lib.lib.sf_debug_precise_attention(precise)
        eng = speaker.SpeakerEngine(enc, dec, store)
        eng.persistent = persistent
        eng.teacher_batched = False          # (this tool reads the word loops' own tapes)
        with torch.no_grad():
            st = eng.score(batch, n, feedback, train=False)
        torch.cuda.synchronize()
        assert st.persistent == persistent
        ok = np.array_equal(st.words[1:].cpu().numpy(), words)
        ctx = st.ctx.cpu().numpy().astype(f8)
        h1 = st.tape['h1'].cpu().numpy().astype(f8)
        c1 = st.tape['c1'].cpu().numpy().astype(f8)
        lg = st.logits.cpu().numpy().astype(f8)
        hin = st.h_init.cpu().numpy().astype(f8)
        cin = st.c_init.cpu().numpy().astype(f8)
        print('== %s (%s): words equal golden %s; ctx dist %.2e, h_init %.2e, c_init %.2e' % (
            name, feedback, ok, np.abs(ctx - ctx64).max(), np.abs(hin - h064).max(), np.abs(cin - c064).max()))
        # ---- the path encoder, step by step: local error of each stage given HIP's own inputs to it
        W = senc_w
        pv = 'visual_attention_layer.'
        xin = st.e['xin'].cpu().numpy().astype(f8)
        al = st.e['alpha'].cpu().numpy().astype(f8)
        hs = st.e['hs'].cpu().numpy().astype(f8)
        cs = st.e['cs'].cpu().numpy().astype(f8)
        F = xin.shape[2] // 2
        print('  path step | alpha (from HIP h) | feature (from HIP alpha) | h c (from HIP xin, h, c: gate product + cell) | trajectory h')
        h64 = np.zeros_like(hs[0])
        c64 = np.zeros_like(hs[0])
        for t in range(len(acts)):
            X = feats[t].astype(f8)
            tv = hs[t] @ W[pv + 'linear_in_h.weight'].astype(f8).T + W[pv + 'linear_in_h.bias']
            q = tv @ W[pv + 'linear_in_v.weight'].astype(f8)
            att = np.einsum('bvf,bf->bv', X, q)
            att = att - att.max(1, keepdims=True)
            e = np.exp(att)
            a = e / e.sum(1, keepdims=True)
            feat_from_alpha = np.einsum('bv,bvf->bf', al[t], X)
            gates = xin[t] @ W['lstm.weight_ih'].astype(f8).T + W['lstm.bias_ih'] + hs[t] @ W['lstm.weight_hh'].astype(f8).T + W['lstm.bias_hh']
            Hh = 512
            i_, f_, g_, o_ = sig(gates[:, :Hh]), sig(gates[:, Hh:2 * Hh]), np.tanh(gates[:, 2 * Hh:3 * Hh]), sig(gates[:, 3 * Hh:])
            c_loc = f_ * cs[t] + i_ * g_
            h_loc = o_ * np.tanh(c_loc)
            print('  %d | %.2e | %.2e (|feat| %.2f) | %.2e %.2e (|gates| %.1f) | %.2e' % (
                t, np.abs(al[t] - a).max(), np.abs(xin[t][:, F:] - feat_from_alpha).max(), np.abs(feat_from_alpha).max(),
                np.abs(hs[t + 1] - h_loc).max(), np.abs(cs[t + 1] - c_loc).max(), np.abs(gates).max(),
                np.abs(hs[t + 1] - ctx64[:, t]).max()))
        print('  t | trajectory: h1 c1 logit | local: h1 c1 logit (logit from HIP h1_t: head only)')
        w = np.full(100, 3, np.int64)
        hp, cp = hin, cin
        tl = []
        for t in range(n):
            loc = dec_step(sdec_w, w, hp, cp, ctx, path_mask)
            # head only: exact attention + projection from HIP's own h1_t
            H = 512
            tq = h1[t] @ sdec_w['attention_layer.linear_in.weight'].astype(f8).T
            att = np.where(path_mask, -np.inf, np.einsum('bld,bd->bl', ctx, tq))
            att = att - att.max(1, keepdims=True)
            e = np.exp(att)
            a = e / e.sum(1, keepdims=True)
            wc = np.einsum('bl,bld->bd', a, ctx)
            ht = np.tanh(np.concatenate((wc, h1[t]), 1) @ sdec_w['attention_layer.linear_out.weight'].astype(f8).T)
            lh = ht @ sdec_w['decoder2action.weight'].astype(f8).T + sdec_w['decoder2action.bias']
            row = (np.abs(h1[t] - traj[t]['h1']).max(), np.abs(c1[t] - traj[t]['c1']).max(), np.abs(lg[t] - traj[t]['logit']).max(),
                   np.abs(h1[t] - loc['h1']).max(), np.abs(c1[t] - loc['c1']).max(), np.abs(lg[t] - lh).max())
            tl.append(row)
            if t < 3 or t % 8 == 7 or t == n - 1:
                print(' %2d | %.2e %.2e %.2e | %.2e %.2e %.2e' % ((t,) + row))
            hp, cp = h1[t], c1[t]
            w = words[t]
        tl = np.array(tl)
        print('  max over steps: trajectory h1 %.2e c1 %.2e logit %.2e | local h1 %.2e c1 %.2e logit-head %.2e' % tuple(tl.max(0)))
    _lib.lib.sf_debug_gate_product_f32(0)
    _lib.lib.sf_debug_precise_attention(1)


if __name__ == '__main__':
    main()
