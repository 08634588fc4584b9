"""The real-R2R fixture world shared by the SR-parity tests: 156 instruction items of the reference's
R2R_sub_train.json on the five committed connectivity graphs (tests/golden/r2r_fixture_items.json),
seeded synthetic features and "peaky" weights (tests/golden/make_golden_r2r.py)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONN = os.path.join(ROOT, 'tests', 'golden', 'connectivity')


def load():
    fx = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'r2r_fixture_items.json')))
    gold = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g10_r2r_eval.json')))
    return fx['items'], gold


def build_env(items, cfg, dense):
    from speaker_follower_amd.build import build_sim
    build_sim(verbose=False)
    from speaker_follower_amd import env, synth
    graphs = {s: env.NavGraph(os.path.join(CONN, s + '_connectivity.json')) for s in cfg['scans']}
    row_of, n = {}, 0
    for s in cfg['scans']:
        for v in graphs[s].ids:
            row_of[s + '_' + v] = n
            n += 1
    table = synth.feature_table(cfg['table_seed'], n)
    its = [dict(it, instr_encoding=np.asarray(it['instr_encoding'], np.int64)) for it in items]
    e = env.R2RIndexEnv(its, row_of, CONN, batch_size=cfg['batch'], host_table=table if dense else None)
    return e, table, graphs


def gt_of(items):
    return {it['path_id']: it for it in items}
