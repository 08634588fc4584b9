#!/usr/bin/env python3
"""Round-2 golden vectors: the cases where parity was soft (run in the BUILD container only).

Like make_golden.py this imports the reference's tasks/R2R/{model,env,follower}.py from
/root/reference on torch-CPU fp32 and stores only OUTPUTS of the reference (plus seeds):

  g8_follower_peaky_b100_argmax   "peaky" weights (synth.follower_weights_peaky: O(1) logit spread,
                                  attention maxima ~0.8), B = 100, argmax feedback, a batch seed for
                                  which all 20 decode steps are live: per-step logits, actions,
                                  visual attention, final h/c, loss
  g8_follower_peaky_b100_train    the same weights, teacher forcing with rare stops (most rows live
                                  for all 20 steps), TRAIN mode: the reference's nn.Dropout modules
                                  are replaced by a module that applies the masks this repo's
                                  counter-based generator produces (oracle/rng.py; seed / sites stored
                                  in the file), so loss and gradients are reproducible: loss, gradient
                                  norms + sampled entries
  g9_speaker_b100_teacher/argmax  speaker at B = 100 ("peaky" speaker weights): 80-step teacher NLL +
                                  gradients, 40 greedy words

    python tests/golden/make_golden_hard.py
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

DROP_SEED = 0x5EED5                  # engine.dropout_seed of the train-mode case
ENC_SEED_XOR = 0x5BD1E995            # FollowerEngine: encoder mask seed = seed ^ this


class MaskedDrop(torch.nn.Module):
    """Stands in for nn.Dropout inside the reference modules: multiplies by the next queued mask
    (already scaled by 1/(1-p)), in call order."""

    def __init__(self):
        super().__init__()
        self.queue = []

    def forward(self, x):
        m = self.queue.pop(0)
        assert m.shape == x.shape, (m.shape, x.shape)
        return x * m


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_model, ref_env, ref_follower = import_reference()
    dims = synth.FULL
    H, F = dims.hidden, dims.feat
    loc_table = np_env.static_loc_embeddings()
    out = {}

    enc_w, dec_w = synth.follower_weights_peaky(303, dims)
    enc = ref_model.EncoderLSTM(dims.vocab, dims.word, dims.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = ref_model.AttnDecoderLSTM(dims.feat, dims.hidden, 0.5, feature_size=dims.feat)
    load(enc, enc_w)
    load(dec, dec_w)

    def rollout(fb, table, steps, feedback, masks=None):
        """follower.py:430-539 over precomputed observations with the REFERENCE modules."""
        seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, 80, reverse=True)
        B = seq.shape[0]
        if masks is not None:                      # train mode with this repo's masks
            T = max(lens)
            enc.drop.queue = [t(masks('ctx', B, T))]
            dec.drop.queue = []
            for st in range(steps):
                dec.drop.queue += [t(m) for m in masks(st, B, T)]
        ctx, h, c = enc(t(seq), lens)
        u_prev = dec.u_begin.expand(B, -1)
        ended = np.zeros(B, bool)
        crit = torch.nn.CrossEntropyLoss(ignore_index=-1)
        loss = 0
        scores = torch.zeros(B)
        logits, actions, alphas_v, alphas, live = [], [], [], [], []
        for st in range(steps):
            X, all_u, is_valid = np_env.dense_follower_step(table, loc_table, fb, st)
            all_u_t = t(all_u)
            h, c, alpha, logit, alpha_v = dec(u_prev, all_u_t, t(X), h, c, ctx, t(mask))
            logit[t(is_valid) == 0] = -float('inf')
            target = t(np.where(ended, -1, fb.target[st]))
            live.append(int((~ended).sum()))
            if (target != -1).any():
                loss = loss + crit(logit, target)
            if feedback == 'teacher':
                a_t = torch.clamp(target, min=0)
            else:
                _, a_t = logit.max(1)
                a_t = a_t.detach()
            u_prev = all_u_t[np.arange(B), a_t, :].detach()
            scores += -torch.nn.functional.cross_entropy(logit, a_t, reduction='none').data
            logits.append(logit.detach().numpy().copy())
            actions.append(a_t.numpy().copy())
            alphas_v.append(alpha_v.detach().numpy().copy())
            alphas.append(alpha.detach().numpy().copy())
            ended |= (a_t.numpy() == 0)
            if ended.all():
                break
        A = fb.a_max
        lg = np.full((len(logits), B, A), -np.inf, np.float32)
        for i, l in enumerate(logits):
            lg[i, :, :l.shape[1]] = l
        res = dict(actions=np.stack(actions), loss=np.float32(float(loss)), scores=scores.numpy(),
                   h=h.detach().numpy(), c=c.detach().numpy(), n_steps=np.int64(len(logits)), logits=lg,
                   alpha_v=np.stack(alphas_v), alpha_last=alphas[-1], live_rows=np.asarray(live, np.int64))
        return res, loss

    # ---- g8 argmax: find a batch seed whose rollout keeps at least one row alive for all 20 steps
    table = synth.feature_table(8, 256)
    chosen = None
    with torch.no_grad():
        for seed in range(40, 80):
            fb = synth.follower_batch(seed=seed, batch=100, steps=20, n_viewpoints=256)
            res, _ = rollout(fb, table, 20, 'argmax')
            print('argmax seed %d: %d steps, live rows per step %s' % (seed, res['n_steps'], res['live_rows'].tolist()))
            if int(res['n_steps']) == 20:
                chosen = seed
                break
    assert chosen is not None
    res['batch_seed'] = np.int64(chosen)
    res['weight_seed'] = np.int64(303)
    res['table_seed'] = np.int64(8)
    fin = res['logits'][np.isfinite(res['logits'])]
    print('logit std %.3f, max |logit| %.3f, mean max visual attention %.3f' %
          (fin.std(), np.abs(fin).max(), res['alpha_v'].max(2).mean()))
    out['g8_follower_peaky_b100_argmax'] = res

    # ---- g8 train: teacher forcing, rare stops, dropout ON with this repo's masks
    fbt = synth.follower_batch(seed=91, batch=100, steps=20, n_viewpoints=256, stop_prob=1.0 / 40.0)
    enc.train()
    dec.train()
    enc.drop = MaskedDrop()
    dec.drop = MaskedDrop()
    site0 = 0                                      # first rollout of a fresh FollowerEngine
    rows = np.arange(100)

    def masks(which, B, T):
        if which == 'ctx':
            return orng.dropout_mask(DROP_SEED ^ ENC_SEED_XOR, site0, rows, T * H, 0.5).reshape(B, T, H)
        st = which
        return (orng.dropout_mask(DROP_SEED, 2 * (site0 + st), rows, 2 * F, 0.5),
                orng.dropout_mask(DROP_SEED, 2 * (site0 + st) + 1, rows, H, 0.5))

    res, loss = rollout(fbt, table, 20, 'teacher', masks=masks)
    assert not dec.drop.queue and not enc.drop.queue
    enc.zero_grad()
    dec.zero_grad()
    loss.backward()
    grng = np.random.default_rng(808)
    res.update({'enc/' + k: v for k, v in grads_summary(enc, grng).items()})
    res.update({'dec/' + k: v for k, v in grads_summary(dec, grng).items()})
    res.update(batch_seed=np.int64(91), weight_seed=np.int64(303), table_seed=np.int64(8),
               dropout_seed=np.int64(DROP_SEED), site0=np.int64(site0))
    for k in ('logits', 'alpha_v'):                # keep the file small: the first 4 steps pin the forward
        res[k] = res[k][:4]
    print('train: %d steps, live rows %s, loss %.5f' % (res['n_steps'], res['live_rows'].tolist(), res['loss']))
    out['g8_follower_peaky_b100_train'] = res

    # ---- g9 speaker at B = 100
    senc_w, sdec_w = synth.speaker_weights_peaky(404, dims)
    senc = ref_model.SpeakerEncoderLSTM(dims.feat, dims.feat, dims.hidden, 0.5)
    sdec = ref_model.SpeakerDecoderLSTM(dims.vocab, dims.word, dims.hidden, 0.5, glove=sdec_w['embedding.weight'])
    load(senc, senc_w)
    load(sdec, sdec_w)

    def ref_speaker(sb, table, steps, feedback, with_grad):
        acts, feats, path_mask = np_env.dense_speaker_inputs(sb, table, loc_table)
        instr_seq, _, _ = np_env.batch_instructions_from_encoded(sb.instr, 80)
        ctx, h, c = senc([t(a) for a in acts], [t(f) for f in feats])
        B = ctx.shape[0]
        w_t = torch.full((B,), 3, dtype=torch.long)
        ended = np.zeros(B, bool)
        loss = 0
        scores = torch.zeros(B)
        words, logits, alphas = [], [], []
        for st in range(steps):
            h, c, alpha, logit = sdec(w_t.view(-1, 1), h, c, ctx, t(path_mask))
            target = t(instr_seq[:, st]).contiguous()
            if feedback == 'teacher':
                w_t = target
            else:
                _, w_t = logit.max(1)
                w_t = w_t.detach()
            logp = torch.nn.functional.log_softmax(logit, dim=1)
            scores += -torch.nn.functional.nll_loss(logp, w_t, ignore_index=0, reduction='none').data
            if (target != 0).any():
                loss = loss + torch.nn.functional.nll_loss(logp, target, ignore_index=0)
            logits.append(logit.detach().numpy().copy())
            alphas.append(alpha.detach().numpy().copy())
            words.append(w_t.numpy().copy())
            ended |= (w_t.numpy() == 2)
            if ended.all():
                break
        # (fixtures stay small: every 4th row of ctx, the first and the last step's logits)
        res = dict(words=np.stack(words), loss=np.float32(float(loss)), scores=scores.numpy(),
                   ctx_rows4=ctx.detach().numpy()[::4], h=h.detach().numpy(), c=c.detach().numpy(),
                   logits_first=np.stack(logits[:1]), logit_last=logits[-1], alpha_first=np.stack(alphas[:2]),
                   n_steps=np.int64(len(logits)))
        if with_grad:
            senc.zero_grad()
            sdec.zero_grad()
            loss.backward()
            grng = np.random.default_rng(909)
            res.update({'enc/' + k: v for k, v in grads_summary(senc, grng).items()})
            res.update({'dec/' + k: v for k, v in grads_summary(sdec, grng).items()})
        return res

    sb = synth.speaker_batch(seed=17, batch=100, n_viewpoints=256, min_len=10, max_len=79)
    r = ref_speaker(sb, table, 80, 'teacher', True)
    r.update(batch_seed=np.int64(17), weight_seed=np.int64(404), table_seed=np.int64(8))
    out['g9_speaker_b100_teacher'] = r
    with torch.no_grad():
        r = ref_speaker(sb, table, 40, 'argmax', False)
    r.update(batch_seed=np.int64(17), weight_seed=np.int64(404), table_seed=np.int64(8))
    r.pop('ctx_rows4')
    out['g9_speaker_b100_argmax'] = r

    for name, arrays in out.items():
        path = os.path.join(HERE, name + '.npz')
        with tempfile.NamedTemporaryFile(dir=HERE, suffix='.npz', delete=False) as f:
            np.savez_compressed(f, **arrays)
        os.replace(f.name, path)
        print('%-34s %8.1f KB  %d arrays' % (name, os.path.getsize(path) / 1024, len(arrays)))


if __name__ == '__main__':
    main()
