"""GPU: end to end on REAL Matterport connectivity graphs (committed fixtures): the C++ navigation
simulator, the cached panorama sweep, the shortest-path teacher and the HIP path together.
Features and instructions are synthetic (the ResNet TSV and the R2R splits are not available
offline); the graphs, the candidate geometry and the teacher are real."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONN = os.path.join(ROOT, 'tests', 'golden', 'connectivity')
SCANS = ['YmJkqBEsHnH', 'gZ6f7yhEvPG', 'GdvgFV5R1Z5']


@pytest.fixture(scope='module')
def world():
    from speaker_follower_amd.build import build_sim
    build_sim()
    from speaker_follower_amd import env, model, features, follower, agents, synth
    graphs = {s: env.NavGraph(os.path.join(CONN, s + '_connectivity.json')) for s in SCANS}
    rng = np.random.default_rng(1)
    items = env.random_items(graphs, 32, rng, min_len=4, max_len=20)
    row_of, n = {}, 0
    for s, g in graphs.items():
        for v in g.ids:
            row_of[s + '_' + v] = n
            n += 1
    table = synth.feature_table(11, n)
    e = env.R2RIndexEnv(items, row_of, CONN, batch_size=8, host_table=table)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights(101)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    store = features.FeatureStore(table)
    return e, enc, dec, store, follower, agents


def test_agent_on_real_graphs_matches_engine_on_gold_index_batch(world):
    e, enc, dec, store, follower, agents = world
    e.reset_epoch()
    fb, path_obs, path_actions = e.gold_index_batch(10)
    S = fb.vp.shape[0]
    engine = follower.FollowerEngine(enc, dec, store)
    with torch.no_grad():
        st = engine.rollout(follower.DeviceFollowerBatch.from_synth(fb), S, 'teacher', train=False)
    agent = agents.Seq2SeqAgent(e, '/tmp/sf_e2e.json', enc, dec, episode_len=10)
    with torch.no_grad():
        traj, loss = agent._score_obs_actions_and_instructions(
            path_obs, path_actions, [po[0]['instr_encoding'] for po in path_obs])
    np.testing.assert_allclose(float(loss), float(st.loss), rtol=1e-5)
    for tr, pa in zip(traj, path_actions):
        assert tr['actions'] == pa


def test_greedy_agent_walks_real_graph_and_stays_on_it(world):
    e, enc, dec, store, follower, agents = world
    agent = agents.Seq2SeqAgent(e, '/tmp/sf_e2e.json', enc, dec, episode_len=8)
    res = agent.test(use_dropout=False, feedback='argmax')
    assert len(res) == 32
    for r in res.values():
        vps = [p[0] for p in r['trajectory']]
        scan = next(it['scan'] for it in e.data if it['instr_id'] == r['instr_id'])
        g = e.graphs[scan]
        for a, b in zip(vps[:-1], vps[1:]):
            assert a == b or b in g.adj[a]                              # every move is a graph edge


def test_teacher_forced_training_on_real_graphs_reduces_loss(world):
    e, enc, dec, store, follower, agents = world
    import copy
    enc2, dec2 = copy.deepcopy(enc), copy.deepcopy(dec)
    e.reset_epoch()
    fb, _, _ = e.gold_index_batch(10)
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    engine = follower.FollowerEngine(enc2, dec2, store)
    opt = torch.optim.Adam([p for m in (enc2, dec2) for p in m.parameters() if p.requires_grad], lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=False)
        st = engine.rollout(batch, fb.vp.shape[0], 'teacher', train=False)
        st.loss.backward()
        opt.step()
        losses.append(float(st.loss.detach()))
    assert losses[-1] < 0.7 * losses[0], losses


def test_training_with_fused_adam_follows_torch_adam(world):
    """The same six teacher-forced iterations with optim.FusedAdam and with torch.optim.Adam: equal
    losses.  (FusedAdam updates the weights behind torch's back; if the caches keyed on
    `param._version` -- transposed weights, the token table of the encoder -- were not refreshed the
    second iteration would already run on stale copies and the curves would part.)"""
    e, enc, dec, store, follower, agents = world
    import copy
    from speaker_follower_amd import optim
    e.reset_epoch()
    fb, _, _ = e.gold_index_batch(10)
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    curves = []
    for make in (lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=5e-4),
                 lambda ps: optim.FusedAdam(ps, lr=1e-3, weight_decay=5e-4)):
        enc2, dec2 = copy.deepcopy(enc), copy.deepcopy(dec)
        engine = follower.FollowerEngine(enc2, dec2, store)
        opt = make([p for m in (enc2, dec2) for p in m.parameters() if p.requires_grad])
        losses = []
        for _ in range(6):
            opt.zero_grad(set_to_none=False)
            st = engine.rollout(batch, fb.vp.shape[0], 'teacher', train=False)
            st.loss.backward()
            opt.step()
            losses.append(float(st.loss.detach()))
        curves.append(losses)
    assert curves[1][-1] < 0.7 * curves[1][0], curves
    np.testing.assert_allclose(curves[1], curves[0], rtol=2e-4)
