"""Why a best-first search can walk a different (equally valid) order under another fp32 summation order: for the
B = 64, K = 40 state-factored search of tests/test_gpu_search.py, prints per instruction the smallest score gap between
an expanded state and the runner-up of its frontier, with the gate product on the fp32 MFMA and on the bf16 matrix
cores (error-free splitting), and where the two traversals part."""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np                                              # noqa: E402
import torch                                                    # noqa: E402
import search_world as W                                        # noqa: E402
from speaker_follower_amd import model, features, agents, synth, _lib      # noqa: E402

with gzip.open(os.path.join(ROOT, 'tests', 'golden', 'g7_search_b64_k40.json.gz'), 'rt') as f:
    gold = json.load(f)['results']
runs = {}
for f32 in (1, 0):
    _lib.lib.sf_debug_gate_product_f32(f32)
    env, table = W.build_world(dense=True, n_items=W.BIG_ITEMS, batch=W.BIG_BATCH, item_seed=W.BIG_ITEM_SEED)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(W.BIG_FOLLOWER_SEED)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    agent = agents.Seq2SeqAgent(env, '/tmp/sf_tie.json', enc, dec, episode_len=W.BIG_EPISODE_LEN)
    agent.store = features.FeatureStore(table)
    agent.tie_log = []
    env.set_beam_size(W.BIG_K)
    env.reset_epoch()
    trajs, completed, traversed = agent.state_factored_search(W.BIG_K, 1)
    inst = np.concatenate([x[0] for x in agent.tie_log])
    gap = np.concatenate([x[1] - x[2] for x in agent.tie_log])
    best = np.concatenate([x[1] for x in agent.tie_log])
    trav = [[s.world_state.viewpointId for s in tr] for tr in traversed]
    runs[f32] = (inst, gap, best, trav)
_lib.lib.sf_debug_gate_product_f32(0)
for f32, (inst, gap, best, trav) in runs.items():
    bad = [i for i in range(64) if trav[i] != gold[i]['traversed']]
    print('gate product on %s: %d of 64 traversals differ from the reference: %s' % ('fp32 MFMA' if f32 else 'bf16 x 6', len(bad), bad))
    for i in bad:
        g = gap[inst == i]
        k = int(np.argmin(g))
        print('   instruction %d: smallest pick-vs-runner-up gap %.3e at score %.4f (fp32 ulp there: %.1e); gaps below 1e-5: %d of %d picks'
              % (i, g[k], best[inst == i][k], np.spacing(np.float32(abs(best[inst == i][k]))), int((g < 1e-5).sum()), len(g)))
allgap = runs[0][1]
print('all picks: %d, gaps below 1e-6: %d, below 1e-5: %d, below 1e-4: %d' % (len(allgap), (allgap < 1e-6).sum(), (allgap < 1e-5).sum(), (allgap < 1e-4).sum()))
