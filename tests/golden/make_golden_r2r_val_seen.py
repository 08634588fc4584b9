#!/usr/bin/env python3
"""G10b: val-seen success-rate parity on the split the reference ships (BUILD container only).

BASELINE.json's metric carries "val-seen SR parity" (README.md:169, eval.py:29-139).  Released weights and the ResNet
TSV are not available offline; what CAN be pinned is the whole reference stack on the real split:

  * the reference's `R2RBatch` environment (tasks/R2R/env.py, over this repo's navigation-only simulator binary) on ALL
    782 instructions / 260 paths / 51 scans of tasks/R2R/data/R2R_sub_val_seen.json, tokenised by utils.Tokenizer with
    train_vocab.txt, features served by the reference's own `MeanPooledImageFeatures.get_features` from a seeded
    synthetic table;
  * the reference's `Seq2SeqAgent` (follower.py) walking it greedily -- its `beam_search(1)`, which the reference
    documents as reproducing the greedy rollout (follower.py:150-156; `_rollout_with_loss` itself indexes 0-dim tensors
    and cannot run on current torch);
  * the reference's `Evaluation._score_item / score_results` (eval.py:56-139) scoring the result.

Weights: seeded (synth.follower_weights_peaky) with the action-scoring head's biases and output vector
(`decoder2action.linear_in_h.bias`, `linear_in_a.bias`, `linear_out.{weight,bias}`: 769 numbers) BRIEFLY TRAINED -- Adam,
teacher forcing on the gold routes of 300 instructions of R2R_sub_train.json, the rest of the network frozen, the
reference's modules doing the forward -- so that the agent walks instead of stopping after three steps (G10's seeded
weights: 3.0 steps, SR 6.4 %).  The trained numbers travel in the fixture (3 KB); everything else is a function of seeds.

Writes tests/golden/g10b_val_seen_eval.json.gz: config, the trained head, per instruction the reference's trajectory
(viewpoints, headings), score, smallest top-2 logit margin on the way, the reference Evaluation's per-item numbers and
summary.
"""
import gzip
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, HERE)

from make_golden_env import reference_env, REF                     # noqa: E402
from speaker_follower_amd import synth, nav_data                   # noqa: E402

EPISODE_LEN, WEIGHT_SEED, TABLE_SEED, BATCH = 10, 303, 21, 100
TRAIN_MINIBATCHES, TRAIN_ITERS, TRAIN_LR = 3, 400, 0.02
HEAD = ('decoder2action.linear_in_h.bias', 'decoder2action.linear_in_a.bias', 'decoder2action.linear_out.weight',
        'decoder2action.linear_out.bias')


def _try_cuda(x):
    return x.bool() if torch.is_tensor(x) and x.dtype == torch.uint8 else x


def feature_rows(scans):
    """'scan_viewpoint' -> row over the INCLUDED viewpoints of `scans` (sorted), connectivity-file order: the layout
    tests/r2r_val_seen.build_env uses."""
    geo = nav_data.load_geometry()
    row_of, n = {}, 0
    for s in sorted(scans):
        for v, inc in zip(geo[s]['ids'], geo[s]['included']):
            if inc:
                row_of[s + '_' + v] = n
                n += 1
    return row_of, n


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    t00 = time.time()
    ref_env, ref_utils = reference_env()                              # (cwd is now the reference tree)
    import model as ref_model
    import follower as ref_follower
    torch.Tensor.cuda = lambda self, *a, **k: self                    # follower.py:318-319 call .cuda() unconditionally
    ref_follower.try_cuda = _try_cuda                                 # uint8 masks -> bool (current torch)
    ref_env.try_cuda = _try_cuda
    vocab = ref_utils.read_vocab(os.path.join('tasks', 'R2R', 'data', 'train_vocab.txt'))
    tok = ref_utils.Tokenizer(vocab=vocab)

    train_scans = {it['scan'] for it in ref_utils.load_datasets(['sub_train'])}
    val_scans = {it['scan'] for it in ref_utils.load_datasets(['sub_val_seen'])}
    assert val_scans <= train_scans
    scans = sorted(train_scans)
    row_of, n_rows = feature_rows(scans)
    table = synth.feature_table(TABLE_SEED, n_rows)
    print('%d scans, %d feature rows (%.2f GB), %.0f s' % (len(scans), n_rows, table.nbytes / 2 ** 30, time.time() - t00))

    # the reference's own feature class, filled from the table instead of a TSV (env.py:341-383)
    feats = ref_env.MeanPooledImageFeatures.__new__(ref_env.MeanPooledImageFeatures)
    feats.image_feature_datasets = ['synthetic']
    feats.feature_dim = ref_env.MeanPooledImageFeatures.MEAN_POOLED_DIM
    feats.features = {k: table[r] for k, r in row_of.items()}

    d = synth.FULL
    assert len(vocab) == d.vocab
    enc_w, dec_w = synth.follower_weights_peaky(WEIGHT_SEED)
    enc = ref_model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = ref_model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.eval()
    dec.eval()

    # ------------------------------------------------------------------ brief training of the scoring head
    tr_env = ref_env.R2RBatch([feats], batch_size=BATCH, seed=10, splits=['sub_train'], tokenizer=tok)
    agent = ref_follower.Seq2SeqAgent(tr_env, '', enc, dec, episode_len=EPISODE_LEN)
    cache = []                                                        # per step: (h~ [B,H], U [B,A,F], valid [B,A], target [B])
    grabbed = {}
    hook = dec.decoder2action.register_forward_pre_hook(lambda m, args: grabbed.update(h=args[0].detach(), u=args[1].detach()))
    with torch.no_grad():
        for _ in range(TRAIN_MINIBATCHES):
            path_obs, path_actions, enc_instr = tr_env.gold_obs_actions_and_instructions(EPISODE_LEN)
            seq, seq_mask, seq_lengths, perm = ref_follower.batch_instructions_from_encoded(
                enc_instr, agent.max_instruction_length, reverse=True, sort=True)
            ctx, h_t, c_t = enc(seq, seq_lengths)
            B = len(path_obs)
            u_prev = dec.u_begin.expand(B, -1)
            obs = None
            for t in range(EPISODE_LEN):
                nobs, tgt = [], []
                for pi, si in enumerate(perm):
                    if t < len(path_actions[si]):
                        tgt.append(path_actions[si][t])
                        nobs.append(path_obs[si][t])
                    else:
                        tgt.append(-1)
                        nobs.append(obs[pi])
                obs = nobs
                if all(x < 0 for x in tgt):
                    break
                f_t = agent._feature_variables(obs)[0]
                all_u, is_valid, _ = agent._action_variable(obs)
                h_t, c_t, _, logit, _ = dec(u_prev, all_u, f_t, h_t, c_t, ctx, seq_mask)
                target = torch.LongTensor(tgt)
                cache.append((grabbed['h'], grabbed['u'], is_valid, target))
                u_prev = all_u[np.arange(B), torch.clamp(target, min=0), :].detach()
    hook.remove()
    n_live = sum(int((c[3] >= 0).sum()) for c in cache)
    print('cached %d teacher-forced decisions of %d routes, %.0f s' % (n_live, TRAIN_MINIBATCHES * BATCH, time.time() - t00))
    sc = dec.decoder2action
    with torch.no_grad():
        # frozen: W_h h~ and W_a U (the two big products of model.py:342-352) once
        pre = [(torch.nn.functional.linear(h, sc.linear_in_h.weight), torch.nn.functional.linear(u, sc.linear_in_a.weight), v, t)
               for h, u, v, t in cache]
    params = [sc.linear_in_h.bias, sc.linear_in_a.bias, sc.linear_out.weight, sc.linear_out.bias]
    for p in dec.parameters():
        p.requires_grad_(False)
    for p in params:
        p.requires_grad_(True)
    opt = torch.optim.Adam(params, lr=TRAIN_LR)
    for it in range(TRAIN_ITERS):
        opt.zero_grad()
        loss, hit = 0.0, 0
        for th, ca, valid, target in pre:
            logit = sc.linear_out((th + sc.linear_in_h.bias).unsqueeze(1) * (ca + sc.linear_in_a.bias)).squeeze(2)
            logit = logit.masked_fill(valid == 0, -float('inf'))
            loss = loss + torch.nn.functional.cross_entropy(logit, target, ignore_index=-1, reduction='sum')
            hit += int(((logit.argmax(1) == target) & (target >= 0)).sum())
        loss = loss / n_live
        loss.backward()
        opt.step()
        if it % 100 == 0 or it == TRAIN_ITERS - 1:
            print('  head training %3d: CE %.4f, teacher accuracy %.3f' % (it, float(loss), hit / n_live))
    for p in params:
        p.requires_grad_(False)
    head = {k: dec.state_dict()[k].numpy().astype(np.float32) for k in HEAD}
    del tr_env, cache, pre

    # ------------------------------------------------------------------ the reference agent on the reference env
    env = ref_env.R2RBatch([feats], batch_size=BATCH, seed=10, splits=['sub_val_seen'], tokenizer=tok)
    agent = ref_follower.Seq2SeqAgent(env, '', enc, dec, episode_len=EPISODE_LEN)
    margins, rec = {}, {}
    orig_av = agent._action_variable

    def av(obs):
        out = orig_av(obs)
        rec['ids'], rec['valid'] = [ob['instr_id'] for ob in obs], out[2]
        return out
    agent._action_variable = av
    orig_fwd = dec.forward

    def fwd(*a, **k):
        out = orig_fwd(*a, **k)
        lg = out[3].detach().numpy()
        for i, iid in enumerate(rec['ids']):
            v = np.sort(lg[i][rec['valid'][i] > 0])[::-1]
            gap = float(v[0] - v[1]) if len(v) > 1 else float('inf')
            margins[iid] = min(margins.get(iid, float('inf')), gap)
        return out
    dec.forward = fwd
    env.set_beam_size(1)
    env.reset_epoch()
    results, looped = {}, False
    with torch.no_grad():
        while not looped:
            trajs, _, _ = agent.beam_search(1)
            for beam in trajs:
                r = beam[0]
                if r['instr_id'] in results:
                    looped = True
                else:
                    results[r['instr_id']] = r
    assert len(results) == len(env.data) == 782
    print('reference agent walked %d instructions, %.0f s' % (len(results), time.time() - t00))

    # ------------------------------------------------------------------ the reference's own scoring
    sys.argv = ['eval.py']
    import eval as ref_eval
    ev = ref_eval.Evaluation(['sub_val_seen'])
    summary, _ = ev.score_results(results)
    per_item = {}
    for iid, res in results.items():
        r = ev._score_item(iid, res['trajectory'])
        per_item[iid] = dict(viewpoints=[p[0] for p in res['trajectory']], headings=[float(p[1]) for p in res['trajectory']],
                             elevations=[float(p[2]) for p in res['trajectory']],
                             actions=[int(a) for a in res['actions']],
                             nav_error=float(r.nav_error), oracle_error=float(r.oracle_error), steps=int(r.trajectory_steps),
                             length=float(r.trajectory_length), success=bool(r.success),
                             oracle_success=bool(r.oracle_success), score=float(res['score']),
                             min_margin=float(margins[iid]))
    out = dict(config=dict(split='sub_val_seen', episode_len=EPISODE_LEN, weight_seed=WEIGHT_SEED, table_seed=TABLE_SEED,
                           batch=BATCH, scans=scans, n_rows=n_rows, n_items=len(results),
                           train=dict(split='sub_train', minibatches=TRAIN_MINIBATCHES, iters=TRAIN_ITERS, lr=TRAIN_LR,
                                      decisions=n_live)),
               head={k: v.reshape(-1).tolist() for k, v in head.items()},
               summary={k: float(v) for k, v in summary.items()}, items=per_item)
    path = os.path.join(HERE, 'g10b_val_seen_eval.json.gz')
    with gzip.open(path, 'wt') as f:
        json.dump(out, f)
    ms = np.array([v['min_margin'] for v in per_item.values()])
    print('summary', out['summary'])
    print('smallest top-2 logit margins:', np.sort(ms)[:8], ' items below 1e-3:', int((ms < 1e-3).sum()))
    print('wrote', path, os.path.getsize(path), 'bytes, %.0f s' % (time.time() - t00))


if __name__ == '__main__':
    main()
