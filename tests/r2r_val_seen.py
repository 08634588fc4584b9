"""The reference's R2R_sub_val_seen split as a test world: 260 paths / 782 instructions on 51 scans
(tests/golden/r2r_sub_val_seen_items.json.gz: the split's paths with the token ids utils.Tokenizer + train_vocab.txt
give the instructions; written by tests/golden/make_golden_env.py), over the navigation geometry the package ships
(data/r2r_connectivity.npz -> nav_data.connectivity_dir)."""
import gzip
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_items():
    """One item per instruction, in R2RBatch's order (env.py:672-688: for item in split, for j, instr)."""
    fx = json.load(gzip.open(os.path.join(GOLDEN, 'r2r_sub_val_seen_items.json.gz'), 'rt'))
    items = []
    for p in fx['paths']:
        for j, enc in enumerate(p['instr_encodings']):
            items.append(dict(scan=p['scan'], path_id=p['path_id'], path=p['path'], heading=p['heading'],
                              distance=p['distance'], instr_id='%s_%d' % (p['path_id'], j),
                              instr_encoding=np.asarray(enc, np.int64)))
    return items, fx


def load_env_golden():
    return json.load(gzip.open(os.path.join(GOLDEN, 'g15_env_reference.json.gz'), 'rt'))


def load_sr_golden():
    return json.load(gzip.open(os.path.join(GOLDEN, 'g10b_val_seen_eval.json.gz'), 'rt'))


def build_env(items, batch_size=100, host_table=None, table_seed=None, scans=None):
    """env.R2RIndexEnv over `scans` (default: the items' scans); feature rows = the included viewpoints of those scans,
    scan by scan (sorted), in connectivity-file order.  Returns (env, row_of, n_rows)."""
    from speaker_follower_amd.build import build_sim
    build_sim(verbose=False)
    from speaker_follower_amd import env, nav_data
    scans = sorted(scans if scans is not None else {it['scan'] for it in items})
    conn = nav_data.connectivity_dir(scans=scans)
    geo = nav_data.load_geometry()
    row_of, n = {}, 0
    for s in scans:
        for v, inc in zip(geo[s]['ids'], geo[s]['included']):
            if inc:
                row_of[s + '_' + v] = n
                n += 1
    if table_seed is not None:
        from speaker_follower_amd import synth
        host_table = synth.feature_table(table_seed, n)
    e = env.R2RIndexEnv(list(items), row_of, conn, batch_size=batch_size, seed=10, host_table=host_table, scans=scans)
    return e, row_of, n
