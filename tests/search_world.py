"""Seeded world shared by the search golden generator (tests/golden/make_golden_search.py) and the
search parity tests: real connectivity graphs (committed fixtures), synthetic items / features /
weights.  Everything is a deterministic function of the seeds below, so the golden file only has to
carry the reference's OUTPUTS."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONN = os.path.join(ROOT, 'tests', 'golden', 'connectivity')
SCANS = ['YmJkqBEsHnH', 'gZ6f7yhEvPG', 'GdvgFV5R1Z5']
N_ITEMS, BATCH = 16, 8
FOLLOWER_SEED, SPEAKER_SEED, TABLE_SEED, ITEM_SEED = 101, 202, 11, 5
EPISODE_LEN, INSTRUCTION_LEN = 6, 12


class ListTokenizer:
    """Stand-in for utils.Tokenizer.decode_sentence (utils.py:109-118) without a vocabulary file."""

    def decode_sentence(self, encoding, break_on_eos=False, join=True):
        out = []
        for ix in encoding:
            if ix == (2 if break_on_eos else 0):
                break
            out.append(str(int(ix)))
        return ' '.join(out) if join else out


# BASELINE.json configs[4]: state-factored search, K = 40 completions, batch 64 (rational_follower.py:42-47)
BIG_ITEMS = BIG_BATCH = 64
BIG_K, BIG_EPISODE_LEN, BIG_ITEM_SEED, BIG_FOLLOWER_SEED = 40, 8, 15, 303


def build_world(dense=True, n_items=N_ITEMS, batch=BATCH, item_seed=ITEM_SEED):
    from speaker_follower_amd.build import build_sim
    build_sim()
    from speaker_follower_amd import env, synth
    graphs = {s: env.NavGraph(os.path.join(CONN, s + '_connectivity.json')) for s in SCANS}
    items = env.random_items(graphs, n_items, np.random.default_rng(item_seed), min_len=4, max_len=20)
    row_of, n = {}, 0
    for s, g in graphs.items():
        for v in g.ids:
            row_of[s + '_' + v] = n
            n += 1
    table = synth.feature_table(TABLE_SEED, n)
    e = env.R2RIndexEnv(items, row_of, CONN, batch_size=batch, host_table=table if dense else None)
    e.tokenizer = ListTokenizer()
    return e, table
