#!/usr/bin/env python3
"""Success-rate parity on REAL R2R items (BUILD container only).

BASELINE.json's metric carries "val-seen SR parity".  Trained weights and the ResNet TSV are not
available offline, so the achievable form is: on real R2R items (paths, headings and instructions
from the reference's tasks/R2R/data/R2R_sub_train.json, tokenised by the reference's own
utils.Tokenizer with its train_vocab.txt) over the real connectivity graphs, with the SAME (seeded)
weights and features, the REFERENCE agent's greedy test run and this repo's must produce the same
trajectories -- hence the same navigation error, success rate and oracle rate as computed by the
reference's own eval.py (Evaluation._score_item / score_results, error margin 3 m).

Writes
  tests/golden/r2r_fixture_items.json   the 52 sub_train paths (x3 instructions) that lie on the five
                                        scans committed under tests/golden/connectivity, with their
                                        token ids (data of the reference, no code)
  tests/golden/g10_r2r_eval.json        the reference agent's trajectories for them and the reference
                                        Evaluation's per-item and summary scores
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from make_golden import import_reference, load, REF          # noqa: E402
from make_golden_search import _Featurizer, _shim_try_cuda    # noqa: E402
import search_world as W                                      # noqa: E402
from speaker_follower_amd import synth                        # noqa: E402

EPISODE_LEN, WEIGHT_SEED, TABLE_SEED, BATCH = 10, 303, 12, 39


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    from speaker_follower_amd.build import build_sim
    build_sim()
    from speaker_follower_amd import sim, env
    real_sim = sim.load()
    ref_model, ref_env, ref_follower = import_reference()
    sys.modules['MatterSim'] = real_sim
    torch.Tensor.cuda = lambda self, *a, **k: self
    ref_follower.try_cuda = _shim_try_cuda
    import utils as ref_utils

    scans = sorted(s for s in os.listdir(W.CONN) if s.endswith('_connectivity.json'))
    scans = [s.split('_')[0] for s in scans]
    data = json.load(open(os.path.join(REF, 'tasks', 'R2R', 'data', 'R2R_sub_train.json')))
    vocab = ref_utils.read_vocab(os.path.join(REF, 'tasks', 'R2R', 'data', 'train_vocab.txt'))
    tok = ref_utils.Tokenizer(vocab=vocab)
    items = []
    for it in data:
        if it['scan'] not in scans:
            continue
        for j, instr in enumerate(it['instructions']):
            enc, _ = tok.encode_sentence(instr)
            items.append(dict(scan=it['scan'], path_id=it['path_id'], path=it['path'], heading=it['heading'],
                              distance=it['distance'], instr_id='%d_%d' % (it['path_id'], j),
                              instr_encoding=[int(x) for x in enc]))
    used = sorted({it['scan'] for it in items})
    print('%d instruction items on %d fixture scans %s, vocab %d' % (len(items), len(used), used, len(vocab)))
    with open(os.path.join(HERE, 'r2r_fixture_items.json'), 'w') as f:
        json.dump(dict(source='tasks/R2R/data/R2R_sub_train.json restricted to the fixture scans; tokens by '
                              'utils.Tokenizer with tasks/R2R/data/train_vocab.txt', vocab_size=len(vocab),
                       items=items), f)

    # world: this repo's index env over the real items, synthetic features, "peaky" weights
    graphs = {s: env.NavGraph(os.path.join(W.CONN, s + '_connectivity.json')) for s in used}
    row_of, n = {}, 0
    for s in used:
        for v in graphs[s].ids:
            row_of[s + '_' + v] = n
            n += 1
    table = synth.feature_table(TABLE_SEED, n)
    its = [dict(it, instr_encoding=np.asarray(it['instr_encoding'], np.int64)) for it in items]
    e = env.R2RIndexEnv(its, row_of, W.CONN, batch_size=BATCH, host_table=table)
    e.tokenizer = W.ListTokenizer()
    e.image_features_list = [_Featurizer()]
    d = synth.FULL
    assert len(vocab) == d.vocab
    enc_w, dec_w = synth.follower_weights_peaky(WEIGHT_SEED)
    enc = load(ref_model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight']), enc_w)
    dec = load(ref_model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat), dec_w)
    agent = ref_follower.Seq2SeqAgent(e, '', enc, dec, episode_len=EPISODE_LEN)
    # BaseAgent.test (follower.py:135-192) drives rollout(); the greedy `_rollout_with_loss` cannot run on
    # current torch (`a_t[i].data[0]` on 0-dim tensors, follower.py:510), so the loop is driven over the
    # reference's beam_search(1), which the reference itself documents as reproducing the greedy rollout
    # (follower.py:150-156)
    enc.eval()
    dec.eval()
    e.set_beam_size(1)
    e.reset_epoch()
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
    assert len(results) == len(items)

    # the reference's own scoring (eval.py:56-139) over the reference's graph loader
    cwd = os.getcwd()
    import networkx as nx
    # eval.py imports `train` (argparse side effects, MatterSim env): its two scoring methods need only
    # gt / distances / error_margin / instr_ids, so they are taken from the class without running
    # Evaluation.__init__ (which loads full splits from relative paths)
    os.chdir(REF)
    try:
        sys.argv = ['eval.py']
        import eval as ref_eval
        G = ref_utils.load_nav_graphs(used)
    finally:
        os.chdir(cwd)
    ev = ref_eval.Evaluation.__new__(ref_eval.Evaluation)
    ev.error_margin = 3.0
    ev.splits = ['fixture']
    ev.gt = {it['path_id']: it for it in data if it['scan'] in used}
    ev.instr_ids = {it['instr_id'] for it in items}
    ev.scans = set(used)
    ev.graphs = G
    ev.distances = {s: dict(nx.all_pairs_dijkstra_path_length(g)) for s, g in G.items()}
    summary, scores = ev.score_results(results)
    per_item = {}
    for instr_id, res in results.items():
        r = ev._score_item(instr_id, res['trajectory'])
        per_item[instr_id] = dict(viewpoints=[p[0] for p in res['trajectory']],
                                  headings=[float(p[1]) for p in res['trajectory']],
                                  nav_error=float(r.nav_error), oracle_error=float(r.oracle_error),
                                  steps=int(r.trajectory_steps), length=float(r.trajectory_length),
                                  success=bool(r.success), oracle_success=bool(r.oracle_success),
                                  score=float(res['score']))
    out = dict(config=dict(episode_len=EPISODE_LEN, weight_seed=WEIGHT_SEED, table_seed=TABLE_SEED, batch=BATCH,
                           scans=used, n_items=len(items)),
               summary={k: float(v) for k, v in summary.items()}, items=per_item)
    path = os.path.join(HERE, 'g10_r2r_eval.json')
    with open(path, 'w') as f:
        json.dump(out, f)
    print('summary', out['summary'])
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
