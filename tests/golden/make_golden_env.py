#!/usr/bin/env python3
"""G15: the reference's OWN environment code pinned -- BUILD container only.

Every earlier golden that involves an environment (G7, G10, G13) was produced by the reference AGENT walking this
repo's `env.R2RIndexEnv`; a candidate-ordering or tie difference between `R2RIndexEnv` and the reference's `R2RBatch`
would be invisible to all of them.  Here the reference's tasks/R2R/env.py itself is imported from /root/reference and
run -- `EnvBatch` / `R2RBatch`, `_get_panorama_states` (env.py:149-224), `_navigate_to_location` (:126-146),
`_shortest_path_action` (:742-761), `observe` (:763-804), `_next_minibatch` (:723-735),
`shortest_paths_to_goals` / `gold_obs_actions_and_instructions` (:823-854), with networkx's all-pairs Dijkstra as the
planner (utils.py:26-51) -- on the REAL R2R_sub_val_seen.json split and the reference's own connectivity files.  The one
thing underneath it that is not the reference's is the simulator binary: the reference's MatterSim needs OpenCV / GL
and cannot be built here, so `sys.modules['MatterSim']` is this repo's navigation-only simulator (N1, pinned on its own
by the Catch tables in tests/test_mattersim_nav.py).

Writes (data only: inputs and the reference's outputs)
  tests/golden/r2r_sub_val_seen_items.json.gz   the 260 paths / 782 instructions of the split with the token ids
                                                utils.Tokenizer + train_vocab.txt give them
  tests/golden/g15_env_reference.json.gz        (a) the order of the minibatches R2RBatch draws (seed 10), sorted and not
                                                (b) for ~600 (scan, viewpoint, heading) states: viewIndex, the heading the
                                                    simulator snapped to, the adj_loc_list IN ORDER (nextViewpointId,
                                                    absViewIndex, rel_heading, rel_elevation), the teacher action towards
                                                    a goal, and the world state each candidate leads to (env.step)
                                                (c) gold_obs_actions_and_instructions over the whole split: per
                                                    instruction the visited (viewpoint, viewIndex, heading), the teacher
                                                    actions and the length of every adj_loc_list on the way
"""
import gzip
import json
import math
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get('SF_REFERENCE', '/root/reference')
SPLIT = 'sub_val_seen'
BATCH, MAX_STEPS, N_STATES, STATE_SEED = 100, 10, 600, 15


def reference_env():
    """The reference's env / utils modules over this repo's simulator binary, cwd = the reference tree (its loaders use
    the relative paths 'connectivity/...' and 'tasks/R2R/data/...')."""
    from speaker_follower_amd.build import build_sim
    build_sim()
    from speaker_follower_amd import sim
    sys.modules['MatterSim'] = sim.load()
    sys.path.insert(0, os.path.join(REF, 'tasks', 'R2R'))
    os.chdir(REF)
    import env as ref_env
    import utils as ref_utils
    return ref_env, ref_utils


class _TinyFeatures:
    """observe() asks the featurizer for a [36, D] array per state (env.py:771); what is pinned here is geometry, so D = 4."""

    def __init__(self):
        self.f = np.arange(36 * 4, dtype=np.float32).reshape(36, 4)

    def get_features(self, state):
        return self.f


def adj_rows(adj):
    return [[d['nextViewpointId'], int(d['absViewIndex']), float(d.get('rel_heading', 0.0)),
             float(d.get('rel_elevation', 0.0))] for d in adj]


def main():
    ref_env, ref_utils = reference_env()
    vocab = ref_utils.read_vocab(os.path.join('tasks', 'R2R', 'data', 'train_vocab.txt'))
    tok = ref_utils.Tokenizer(vocab=vocab)
    batch = ref_env.R2RBatch([_TinyFeatures()], batch_size=BATCH, seed=10, splits=[SPLIT], tokenizer=tok)
    n_items = len(batch.data)

    # ---- the split as data: paths once, instructions with their token ids (the order of the JSON file)
    raw = ref_utils.load_datasets([SPLIT])
    paths = [dict(path_id=it['path_id'], scan=it['scan'], heading=it['heading'], distance=it['distance'], path=it['path'],
                  instr_encodings=[[int(x) for x in tok.encode_sentence(s)[0]] for s in it['instructions']],
                  instr_lengths=[int(tok.encode_sentence(s)[1]) for s in it['instructions']])
             for it in raw]
    with gzip.open(os.path.join(HERE, 'r2r_%s_items.json.gz' % SPLIT), 'wt') as f:
        json.dump(dict(source='tasks/R2R/data/R2R_%s.json; token ids by utils.Tokenizer with train_vocab.txt '
                              '(instruction text dropped)' % SPLIT, vocab_size=len(vocab), paths=paths), f)

    out = dict(config=dict(split=SPLIT, batch=BATCH, seed=10, max_steps=MAX_STEPS, n_items=n_items,
                           n_states=N_STATES, state_seed=STATE_SEED))

    # ---- (a) minibatch order (env.py:693-694 shuffle, :723-735 draw): one and a half epochs, sorted and unsorted
    order = []
    for k in range(12):
        batch._next_minibatch(k % 2 == 0)
        order.append([it['instr_id'] for it in batch.batch])
    out['minibatches'] = order

    # ---- (b) states: panorama sweep, teacher, step
    rng = np.random.default_rng(STATE_SEED)
    scans = sorted(batch.scans)
    sim = batch.env.sims[0][0]
    states = []
    while len(states) < N_STATES:
        scan = scans[int(rng.integers(len(scans)))]
        nodes = sorted(batch.graphs[scan].nodes())
        vp = nodes[int(rng.integers(len(nodes)))]
        kind = int(rng.integers(4))
        if kind == 0:                                        # exactly on a snapping boundary (15 degrees + k * 30)
            heading = (int(rng.integers(12)) + 0.5) * math.pi / 6
        elif kind == 1:                                      # exactly on a view centre
            heading = int(rng.integers(12)) * math.pi / 6
        else:
            heading = float(rng.uniform(0, 2 * math.pi))
        elevation = [0.0, 0.0, -math.pi / 6, math.pi / 6][int(rng.integers(4))]
        ws = ref_env.WorldState(scan, vp, heading, elevation)
        ref_env.load_world_state(sim, ws)
        state, adj = ref_env._get_panorama_states(sim)
        goal = nodes[int(rng.integers(len(nodes)))]
        teacher = batch._shortest_path_action(state, adj, goal)
        nxt = []
        for a in range(len(adj)):
            ref_env.load_world_state(sim, ws)
            ref_env._navigate_to_location(sim, adj[a]['nextViewpointId'], adj[a]['absViewIndex'])
            w2 = ref_env.get_world_state(sim)
            nxt.append([w2.viewpointId, float(w2.heading), float(w2.elevation), int(sim.getState().viewIndex)])
        states.append(dict(scan=scan, viewpoint=vp, heading=heading, elevation=elevation,
                           viewIndex=int(state.viewIndex), snapped_heading=float(state.heading),
                           snapped_elevation=float(state.elevation), adj=adj_rows(adj), goal=goal, teacher=int(teacher),
                           next=nxt))
    out['states'] = states

    # ---- (c) the gold routes of the whole split (the speaker's training input, speaker.py:376-395)
    batch.reset_epoch()
    random.seed(10)
    routes, seen = {}, 0
    while len(routes) < n_items:
        path_obs, path_actions, enc = batch.gold_obs_actions_and_instructions(MAX_STEPS)
        for obs, acts, e in zip(path_obs, path_actions, enc):
            k = obs[0]['instr_id']
            if k in routes:
                continue
            routes[k] = dict(viewpoints=[ob['viewpoint'] for ob in obs], views=[int(ob['viewIndex']) for ob in obs],
                             headings=[float(ob['heading']) for ob in obs], actions=[int(a) for a in acts],
                             a_num=[len(ob['adj_loc_list']) for ob in obs], n_tokens=len(e))
        seen += 1
        assert seen < 40
    out['routes'] = routes
    lens = [len(r['actions']) for r in routes.values()]
    print('%d items, %d states, %d routes (teacher steps incl. stop: mean %.2f, max %d), %d minibatches'
          % (n_items, len(states), len(routes), np.mean(lens), max(lens), len(order)))
    path = os.path.join(HERE, 'g15_env_reference.json.gz')
    with gzip.open(path, 'wt') as f:
        json.dump(out, f)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
