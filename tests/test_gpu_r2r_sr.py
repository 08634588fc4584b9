"""GPU: success-rate parity on REAL R2R items (BASELINE.json: "val-seen SR parity").

156 instructions of the reference's R2R_sub_train.json on five real connectivity graphs, tokenised by
the reference's Tokenizer; same seeded weights and features on both sides.  The reference agent's
greedy run (its beam_search(1): follower.py:150-156, 541-718) and the reference Evaluation (eval.py)
are pinned in tests/golden/g10_r2r_eval.json; here Seq2SeqAgent.test() on the HIP path -- the per-step
host loop, the device-resident environment and beam_search(1) -- must walk the SAME viewpoints for
every instruction, hence the same navigation errors, success rate (10 / 156) and oracle rate."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import r2r_world                                                    # noqa: E402
from oracle import np_env                                           # noqa: E402  (checker: eval.py restated)


@pytest.fixture(scope='module')
def world():
    from speaker_follower_amd import model, features, agents, synth, nav
    items, gold = r2r_world.load()
    cfg = gold['config']
    env, table, graphs = r2r_world.build_env(items, cfg, dense=True)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(cfg['weight_seed'])
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    agent = agents.Seq2SeqAgent(env, '/tmp/sf_r2r.json', enc, dec, episode_len=cfg['episode_len'])
    agent.store = features.FeatureStore(table)
    return items, gold, env, graphs, agent, nav.NavTable(env, agent.store)


def check(results, items, gold, graphs):
    assert set(results) == set(gold['items'])
    for k, w in gold['items'].items():
        assert [p[0] for p in results[k]['trajectory']] == w['viewpoints'], k
        np.testing.assert_allclose([p[1] for p in results[k]['trajectory']], w['headings'], atol=1e-9)
        assert abs(results[k]['score'] - w['score']) <= 3e-4 * max(1.0, abs(w['score']))
    summary, per_item = np_env.score_results(r2r_world.gt_of(items), graphs, results)
    for k in ('success_rate', 'oracle_rate', 'nav_error', 'oracle_error', 'steps', 'lengths'):
        np.testing.assert_allclose(summary[k], gold['summary'][k], rtol=1e-9)
    assert summary['success_rate'] == gold['summary']['success_rate'] == 10 / 156
    return summary


@pytest.mark.parametrize('mode', ['host_loop', 'device_env', 'beam1'])
def test_success_rate_equals_the_reference_on_real_r2r_items(world, mode):
    items, gold, env, graphs, agent, table = world
    agent.nav_table = None
    if mode == 'device_env':
        agent.use_device_env(table)
    with torch.no_grad():
        if mode == 'beam1':
            env.set_beam_size(1)
            env.reset_epoch()
            results, looped = {}, False
            while not looped:
                trajs, _, _ = agent.beam_search(1)
                for beam in trajs:
                    if beam[0]['instr_id'] in results:
                        looped = True
                    else:
                        results[beam[0]['instr_id']] = beam[0]
        else:
            results = agent.test(use_dropout=False, feedback='argmax')
    s = check(results, items, gold, graphs)
    print(mode, 'success_rate %.4f oracle_rate %.4f nav_error %.3f m' % (s['success_rate'], s['oracle_rate'], s['nav_error']))
