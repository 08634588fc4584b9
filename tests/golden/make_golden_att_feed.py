#!/usr/bin/env python3
"""Golden vectors for SpeakerDecoderLSTM(use_input_att_feed=True) (model.py:475-481, 500-513) -- BUILD container only.

Imports the reference's `model.py`, builds the decoder with the input-attention-feed branch on (no reference script
passes the flag; the module itself is reference code), loads this repo's seeded weights
(`synth.speaker_decoder_att_feed_weights`) and runs THREE chained word steps on B = 6 rows in eval mode: per step h1, c1,
alpha, logit; and the gradients of sum(logit_last * g) + sum(h_last * gh) with respect to every trainable parameter, the
initial state and the context (torch autograd over the reference module) -> tests/golden/g14_speaker_att_feed.npz.
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
from make_golden import import_reference, load     # noqa: E402


def main():
    torch.manual_seed(0)
    ref_model, _, _ = import_reference()
    d = synth.FULL
    w = synth.speaker_decoder_att_feed_weights(77, d)
    dec = ref_model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=w['embedding.weight'], use_input_att_feed=True)
    load(dec, w)
    dec.eval()
    rng = np.random.default_rng(1477)
    B, Tp, H = 6, 5, d.hidden
    ctx = torch.tensor(np.tanh(rng.standard_normal((B, Tp, H))).astype(np.float32), requires_grad=True)
    mask = np.zeros((B, Tp), bool)
    mask[1, 3:] = True
    mask[4, 1:] = True
    h0 = torch.tensor(np.tanh(rng.standard_normal((B, H))).astype(np.float32), requires_grad=True)
    c0 = torch.tensor(rng.standard_normal((B, H)).astype(np.float32), requires_grad=True)
    words = rng.integers(4, d.vocab, size=(3, B))
    out = dict(ctx=ctx.detach().numpy(), mask=mask, h0=h0.detach().numpy(), c0=c0.detach().numpy(), words=words,
               weight_seed=np.int64(77))
    h, c = h0, c0
    for t in range(3):
        h, c, alpha, logit = dec(torch.from_numpy(words[t]).view(-1, 1), h, c, ctx, torch.from_numpy(mask))
        out['h1_%d' % t], out['c1_%d' % t] = h.detach().numpy(), c.detach().numpy()
        out['alpha_%d' % t], out['logit_%d' % t] = alpha.detach().numpy(), logit.detach().numpy()
    g = rng.standard_normal(logit.shape).astype(np.float32)
    gh = rng.standard_normal(h.shape).astype(np.float32)
    out['g_logit'], out['g_h'] = g, gh
    ((logit * torch.from_numpy(g)).sum() + (h * torch.from_numpy(gh)).sum()).backward()
    # parameter gradients as norm + 64 sampled entries (the format tests/test_gpu_hard_parity.py: check_grads reads)
    srng = np.random.default_rng(5)
    for k, p in dec.named_parameters():
        if p.grad is not None:
            flat = p.grad.numpy().ravel()
            idx = np.sort(srng.choice(flat.size, size=min(64, flat.size), replace=False))
            out['dec/gnorm/' + k] = np.float64(np.sqrt(np.sum(flat.astype(np.float64) ** 2)))
            out['dec/gidx/' + k] = idx
            out['dec/gval/' + k] = flat[idx]
    out['d_h0'], out['d_c0'], out['d_ctx'] = h0.grad.numpy(), c0.grad.numpy(), ctx.grad.numpy()
    path = os.path.join(HERE, 'g14_speaker_att_feed.npz')
    with tempfile.NamedTemporaryFile(dir=HERE, suffix='.npz', delete=False) as f:
        np.savez_compressed(f, **out)
    os.replace(f.name, path)
    print(path, '%.1f KB' % (os.path.getsize(path) / 1024), sorted(k for k in out if 'gnorm' in k or k.startswith('d_')))


if __name__ == '__main__':
    main()
