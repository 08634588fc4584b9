#!/usr/bin/env python3
"""Exact-arithmetic anchors for the speaker goldens (run in the BUILD container only).

north_star bounds the logits by 1e-4 ABSOLUTE against the reference's fp32 CPU path.  On the "peaky" speaker case
(G9: |logit| up to 11.4) the reference's OWN fp32 output lies 2.3e-4 from the same reference modules evaluated in
float64 -- its summation order costs that much at this scale -- so no fp32 implementation with another (equally valid)
order can be held to 1e-4 of it.  This script stores the reference MODULES' float64 outputs (tasks/R2R/model.py with
`.double()`, words forced to the fp32 golden's so both runs walk the same sequence), rounded to float32, as the anchor:
the GPU tests require the HIP logits within 1e-4 absolute of the ANCHOR and print the fp32 reference's own distance
beside it.

    python tests/golden/make_golden_f64.py   ->  g9_speaker_b100_f64.npz
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
from oracle import np_env                          # noqa: E402
from make_golden import import_reference, load     # noqa: E402


def main():
    torch.manual_seed(0)
    ref_model, _, _ = import_reference()
    dims = synth.FULL
    loc_table = np_env.static_loc_embeddings()
    senc_w, sdec_w = synth.speaker_weights_peaky(404, dims)
    senc = ref_model.SpeakerEncoderLSTM(dims.feat, dims.feat, dims.hidden, 0.5)
    sdec = ref_model.SpeakerDecoderLSTM(dims.vocab, dims.word, dims.hidden, 0.5, glove=sdec_w['embedding.weight'])
    load(senc, senc_w)
    load(sdec, sdec_w)
    senc.double().eval()
    sdec.double().eval()
    torch.set_default_dtype(torch.float64)         # the reference's init_state builds its zeros with the default dtype
    table = synth.feature_table(8, 256)
    sb = synth.speaker_batch(seed=17, batch=100, n_viewpoints=256, min_len=10, max_len=79)
    acts, feats, path_mask = np_env.dense_speaker_inputs(sb, table, loc_table)
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()          # noqa: E731
    out = {}
    with torch.no_grad():
        ctx, h0, c0 = senc([td(a) for a in acts], [td(f) for f in feats])
        for feedback in ('teacher', 'argmax'):
            with np.load(os.path.join(HERE, 'g9_speaker_b100_%s.npz' % feedback)) as g:
                words, n = g['words'], int(g['n_steps'])
                first32, last32 = g['logits_first'][0], g['logit_last']
            h, c = h0, c0
            w_t = torch.full((100,), 3, dtype=torch.long)
            logits = []
            for st in range(n):
                h, c, alpha, logit = sdec(w_t.view(-1, 1), h, c, ctx, torch.from_numpy(path_mask).bool())
                logits.append(logit.numpy().copy())
                w_t = torch.from_numpy(words[st])                   # the fp32 golden's sequence, both modes
            # every word step at 8 fixed vocabulary columns (round 5: the bound is asserted along the whole pass)
            cols = np.array([2, 57, 130, 333, 512, 700, 871, 990])
            out['cols'] = cols
            out[feedback + '/logits_cols'] = np.stack([l[:, cols] for l in logits]).astype(np.float32)
            out[feedback + '/logits_first'] = logits[0].astype(np.float32)
            out[feedback + '/logit_last'] = logits[-1].astype(np.float32)
            print('%-8s fp32 reference vs its float64 evaluation: step 0 %.3e, step %d %.3e (max|logit| %.2f)' % (
                feedback, np.abs(first32 - logits[0]).max(), n - 1, np.abs(last32 - logits[-1]).max(), np.abs(logits[0]).max()))
            out[feedback + '/ref32_dist_first'] = np.float64(np.abs(first32 - logits[0]).max())
            out[feedback + '/ref32_dist_last'] = np.float64(np.abs(last32 - logits[-1]).max())
    path = os.path.join(HERE, 'g9_speaker_b100_f64.npz')
    with tempfile.NamedTemporaryFile(dir=HERE, suffix='.npz', delete=False) as f:
        np.savez_compressed(f, **out)
    os.replace(f.name, path)
    print('%s %.1f KB' % (path, os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
