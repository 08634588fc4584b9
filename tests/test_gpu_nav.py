"""GPU: device-resident navigation (speaker_follower_amd/nav.py, csrc/sf_nav.hip) -- a student-forced
rollout on REAL connectivity graphs with one host sync -- against `agents._rollout_with_loss`, which
mirrors the reference loop (follower.py:430-539: decoder step, D2H of the actions, env.step,
env.observe in Python every step): identical trajectories, actions, teacher targets and loss for
argmax, teacher and sample feedback; replayable as a hipGraph."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import search_world as W          # noqa: E402

EPISODE = 7


@pytest.fixture(scope='module')
def world():
    from speaker_follower_amd import model, features, agents, synth, nav, follower
    env, table = W.build_world(dense=True, n_items=24, batch=12, item_seed=77)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(303)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    agent = agents.Seq2SeqAgent(env, '/tmp/sf_nav.json', enc, dec, episode_len=EPISODE)
    store = features.FeatureStore(table)
    agent.store = store
    table_nav = nav.NavTable(env, store)
    return env, agent, store, table_nav, enc, dec


def test_nav_table_reproduces_the_panorama_sweep(world):
    """Every (viewpoint, view) state: the tabulated candidates equal env.panorama (env.py:149-224)."""
    from speaker_follower_amd.env import WorldState, ANGLE_INC
    env, agent, store, nt, enc, dec = world
    a_num = nt.a_num.cpu().numpy()
    nxt, cv = nt.next_row.cpu().numpy(), nt.cand_view.cpu().numpy()
    sc = nt.sincos.cpu().numpy()
    assert nt.n_rows == sum(sum(g.included) for g in env.graphs.values())
    for r, (scan, vp) in enumerate(nt.vp_of):
        for view in (0, 13, 35):
            _, adj = env.panorama(WorldState(scan, vp, (view % 12) * ANGLE_INC, (view // 12 - 1) * ANGLE_INC))
            s = r * 36 + view
            assert a_num[s] == len(adj) and nxt[s, 0] == r
            for a, d in enumerate(adj[1:], 1):
                assert nt.vp_of[nxt[s, a]] == (scan, d['nextViewpointId']) and cv[s, a] == d['absViewIndex']
                np.testing.assert_allclose(sc[s, a], [np.sin(d['rel_heading']), np.cos(d['rel_heading']),
                                                      np.sin(d['rel_elevation']), np.cos(d['rel_elevation'])],
                                           rtol=1e-6, atol=1e-6)


def _host_rollout(env, agent, feedback):
    env.reset_epoch()
    agent.feedback = feedback
    agent._sample_count = 0
    with torch.no_grad():
        traj = agent._rollout_with_loss()
    return traj, float(agent.loss), list(env.batch)


@pytest.mark.parametrize('feedback', ['argmax', 'teacher', 'sample'])
def test_device_rollout_equals_the_per_step_host_loop(world, feedback):
    from speaker_follower_amd import follower, nav
    env, agent, store, nt, enc, dec = world
    want, want_loss, items = _host_rollout(env, agent, feedback)
    eng = follower.FollowerEngine(enc, dec, store)
    # same counter-based sampling streams as the agent's glue calls (seed, 1-based step counter)
    eng.dropout_seed = agent._sample_seed ^ 0x1B873593
    eng.site_next = 1
    navb = nav.DeviceNavBatch(nt, items, EPISODE)
    with torch.no_grad():
        st = eng.rollout(navb, EPISODE, feedback, train=False)
    got = navb.trajectories(st)
    assert [g['instr_id'] for g in got] == [w['instr_id'] for w in want]
    moved = 0
    for g, w in zip(got, want):
        assert g['actions'] == [int(a) for a in w['actions']], (g['instr_id'], feedback)
        assert len(g['trajectory']) == len(w['trajectory'])
        for pg, pw in zip(g['trajectory'], w['trajectory']):
            assert pg[0] == pw[0] and pg[1] == pytest.approx(pw[1], abs=1e-12) and pg[2] == pytest.approx(pw[2], abs=1e-12)
        np.testing.assert_allclose(g['scores'], w['scores'], rtol=2e-4, atol=2e-4)
        moved += len({p[0] for p in g['trajectory']}) > 1
    assert moved >= len(got) // 2                      # the agents really walk the graph
    np.testing.assert_allclose(float(st.loss), want_loss, rtol=1e-4)
    # the teacher the device derived (shortest-path next hop) is the env's (env.py:742-761)
    tgt = navb.target[:EPISODE].cpu().numpy()
    ws = env.reset(sort=True, load_next_minibatch=False)
    obs = env.observe(ws)
    assert [int(x) for x in tgt[0]] == [ob['teacher'] for ob in obs]


@pytest.mark.parametrize('feedback', ['argmax', 'sample'])
def test_env_step_schedules_agree_bit_for_bit(world, feedback):
    """The schedules of a device-environment rollout -- the whole decode loop as ONE library call
    (sf_follower_episode_fwd with glue.nav: default), the same launches issued call by call from the host loop
    (attention of step t+1 deferred behind the env step, the env step inside the scoring + glue launch), the same
    with a separate sf_nav_step launch, and the plain per-step order -- produce identical states, actions and
    logits."""
    from speaker_follower_amd import follower, nav
    env, agent, store, nt, enc, dec = world
    env.reset_epoch()
    items = list(env.batch)
    outs = []
    for fused, pipelined, episode in ((True, True, True), (True, True, False), (False, True, True), (False, False, True)):
        eng = follower.FollowerEngine(enc, dec, store)
        eng.dropout_seed, eng.site_next = 12345, 1
        eng.fold_text = False           # (the folded text stage exists in the one-call episode only and re-associates:
        #                                 tests/test_gpu_text_fold.py holds it to 3e-5; here: SCHEDULES, bit for bit)
        eng.fused_env_step, eng.pipelined, eng.episode_call = fused, pipelined, episode
        navb = nav.DeviceNavBatch(nt, items, EPISODE)
        with torch.no_grad():
            st = eng.rollout(navb, EPISODE, feedback, train=False)
        torch.cuda.synchronize()
        assert (st.episode is not None) == (fused and pipelined and episode)
        outs.append((st.actions.clone(), st.logits.clone(), navb.row.clone(), navb.view.clone(), navb.target.clone(),
                     navb.cand_view.clone(), navb.sincos.clone(), st.loss.clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_device_rollout_trains_and_replays_as_a_graph(world):
    """BPTT through a device-env rollout gives the gradients of the same rollout fed from the host
    (its recorded index-form observations), and the whole thing is hipGraph-capturable."""
    from speaker_follower_amd import follower, nav, synth
    env, agent, store, nt, enc, dec = world
    env.reset_epoch()
    env._next_minibatch(True)
    items = list(env.batch)
    navb = nav.DeviceNavBatch(nt, items, EPISODE)
    eng = follower.FollowerEngine(enc, dec, store)
    for m in (enc, dec):
        m.zero_grad(set_to_none=True)
    st = eng.rollout(navb, EPISODE, 'argmax', train=False)
    st.loss.backward()
    torch.cuda.synchronize()
    g_nav = {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None}
    # the same observations as a plain host-fed batch (pipelined episode path)
    fb = synth.FollowerBatch(instr=[it['instr_encoding'] for it in items],
                             vp=navb.vp[:EPISODE].cpu().numpy(), view=navb.view[:EPISODE].cpu().numpy(),
                             a_num=navb.a_num[:EPISODE].cpu().numpy(),
                             cand_view=navb.cand_view[:EPISODE].cpu().numpy(),
                             cand_heading=np.zeros((EPISODE, len(items), nt.A), np.float32),
                             cand_elevation=np.zeros((EPISODE, len(items), nt.A), np.float32),
                             target=navb.target[:EPISODE].cpu().numpy(), a_max=nt.A)
    hb = follower.DeviceFollowerBatch.from_synth(fb)
    hb.sincos = navb.sincos[:EPISODE].clone()
    for m in (enc, dec):
        m.zero_grad(set_to_none=True)
    st2 = follower.FollowerEngine(enc, dec, store).rollout(hb, EPISODE, 'argmax', train=False)
    st2.loss.backward()
    assert torch.equal(st2.actions, st.actions)
    np.testing.assert_allclose(float(st2.loss), float(st.loss), rtol=1e-5)
    for k, p in dec.named_parameters():
        if p.grad is not None and float(g_nav[k].abs().max()) > 1e-6:
            np.testing.assert_allclose(p.grad.cpu().numpy(), g_nav[k].cpu().numpy(), rtol=2e-3,
                                       atol=1e-5 * float(g_nav[k].abs().max()), err_msg=k)
    for m in (enc, dec):
        m.zero_grad(set_to_none=True)
    # hipGraph: capture once, replay: same walk
    replay, gst = eng.capture(navb, EPISODE, 'argmax')
    navb.row.zero_()
    replay()
    torch.cuda.synchronize()
    assert torch.equal(gst.actions, st.actions)
    assert [t['trajectory'] for t in navb.trajectories(gst)] == [t['trajectory'] for t in navb.trajectories(st)]


def test_training_graph_over_the_device_environment(world):
    """The reference's own training configuration -- `sample` feedback, the environment stepped ON THE DEVICE by the action
    just drawn -- as hipGraph replays (runtime.TrainingGraph): the sampled walks, losses and weights of the eager loop."""
    from speaker_follower_amd import follower, nav, optim, model, synth
    env, agent, store, nt, enc0, dec0 = world
    env.reset_epoch()
    env._next_minibatch(True)
    items = list(env.batch)
    d = synth.FULL
    out = {}
    for mode in ('eager', 'graph'):
        enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc0.embedding.weight.detach().cpu().numpy())
        dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
        enc.load_state_dict(enc0.state_dict())
        dec.load_state_dict(dec0.state_dict())
        enc.cuda().train()
        dec.cuda().train()
        navb = nav.DeviceNavBatch(nt, items, EPISODE)
        oe = optim.FusedAdam([p for p in enc.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-4)
        od = optim.FusedAdam([p for p in dec.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-4)
        eng = follower.FollowerEngine(enc, dec, store)
        eng.dropout_seed = 31
        losses, walks = [], []

        def note(st, loss):
            torch.cuda.synchronize()
            losses.append(float(loss))
            walks.append([t['trajectory'] for t in navb.trajectories(st)])
        if mode == 'eager':
            for _ in range(4):
                oe.zero_grad()
                od.zero_grad()
                st = eng.rollout(navb, EPISODE, 'sample', train=True)
                st.loss.backward()
                oe.step()
                od.step()
                note(st, st.loss.detach())
        else:
            tg = eng.capture_training(navb, EPISODE, 'sample', optimizers=(oe, od))
            # (the eager first iteration's walk was overwritten by the capture's buffers: compare from replay 1 on)
            losses.append(float(tg.first.loss_buf))
            walks.append(None)
            for _ in range(3):
                st = tg.replay()
                note(st, st.loss_buf)
        torch.cuda.synchronize()
        out[mode] = (losses, walks, torch.cat([p.detach().reshape(-1) for m in (enc, dec) for p in m.parameters()]).clone())
    np.testing.assert_allclose(out['graph'][0], out['eager'][0], rtol=2e-6)
    assert len(set(out['eager'][0])) == 4
    for a, b in list(zip(out['eager'][1], out['graph'][1]))[1:]:
        assert a == b                                               # the same sampled walks through the graphs
    assert out['eager'][1][1] != out['eager'][1][2]                 # (and they differ from iteration to iteration)
    we, wg = out['eager'][2], out['graph'][2]
    assert float((we - wg).abs().max()) <= 2e-6 * float(we.abs().max())


def test_agent_api_with_device_env_trains_and_tests(world, tmp_path):
    """Seq2SeqAgent.train / test (follower.py:987-1020) over the device-resident env: same walk as the
    host loop in eval mode, finite falling loss over a few Adam iterations in train mode."""
    from speaker_follower_amd import agents, optim, synth, model
    env, agent0, store, nt, _, _ = world
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(303)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda()
    dec.cuda()
    want, _, _ = _host_rollout(env, agent0, 'argmax')
    agent = agents.Seq2SeqAgent(env, str(tmp_path / 'r.json'), enc, dec, episode_len=EPISODE)
    agent.store = store
    agent.use_device_env(nt)
    env.reset_epoch()
    res = agent.test(use_dropout=False, feedback='argmax')
    for w in want:
        assert [p[0] for p in res[w['instr_id']]['trajectory']] == [p[0] for p in w['trajectory']]
    env.reset_epoch()
    oe = optim.FusedAdam([p for p in enc.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
    od = optim.FusedAdam([p for p in dec.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
    agent.train(oe, od, 6, feedback='teacher')
    assert len(agent.losses) == 6 and all(np.isfinite(agent.losses))
    assert min(agent.losses[3:]) < agent.losses[0]


def _fresh_agent(world, graph, lr=1e-3, torch_adam=False):
    from speaker_follower_amd import agents, model, optim, synth
    env, _, store, nt, enc0, dec0 = world
    d = synth.FULL
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc0.embedding.weight.detach().cpu().numpy())
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict(enc0.state_dict())
    dec.load_state_dict(dec0.state_dict())
    enc.cuda()
    dec.cuda()
    torch.manual_seed(4)
    ag = agents.Seq2SeqAgent(env, '/tmp/sf_nav_train.json', enc, dec, episode_len=EPISODE)
    ag.store = store
    ag.use_device_env(nt)
    ag.train_graph = graph
    Adam = torch.optim.Adam if torch_adam else optim.FusedAdam
    oe = Adam([p for p in enc.parameters() if p.requires_grad], lr=lr, weight_decay=5e-4)
    od = Adam([p for p in dec.parameters() if p.requires_grad], lr=lr, weight_decay=5e-4)
    # (the env reshuffles its items with `random` when an epoch wraps, env.py:601-614: same order, same generator state
    # for every agent built here)
    import random
    if not hasattr(env, '_items_in_order'):
        env._items_in_order = list(env.data)
    env.data[:] = env._items_in_order
    random.seed(11)
    env.reset_epoch()
    return ag, oe, od, lambda: torch.cat([p.detach().reshape(-1) for m in (enc, dec) for p in m.parameters()]).clone()


def test_agent_train_runs_whole_iterations_as_graph_replays(world):
    """Seq2SeqAgent.train (follower.py:1001-1020) with optim.FusedAdam on the device environment: after the first
    iteration every iteration is one replay over the NEXT minibatch (nav.DeviceNavBatch.load) -- the losses and the
    weights of the loop that issues every launch (instructions padded to the minibatch's longest there, to
    max_instruction_length here: equal up to the summation order of the padded attention columns)."""
    out = {}
    for graph in (False, True):
        # (teacher feedback, dropout 0.5, lr 1e-4: a drawn action would turn a 1e-7 difference in a probability into a
        # different walk sooner or later; that a replay draws the eager loop's samples is pinned bit for bit at engine
        # level, test_training_graph_over_the_device_environment)
        ag, oe, od, weights = _fresh_agent(world, graph, lr=1e-4)
        ag.train(oe, od, 3, feedback='teacher')
        first = list(ag.losses)
        ag.train(oe, od, 3, feedback='teacher')                # (a second call: the graph is kept)
        assert (ag.__dict__.get('_train_graph_state') is not None) == graph
        if graph:
            assert ag._train_graph_state[1].replays == 5 and oe.host_steps() == [6] and od.host_steps() == [6]
        out[graph] = (first + list(ag.losses), weights())
    print('[agent.train] losses eager', out[False][0], 'graph', out[True][0])
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-4)
    assert len(set(out[True][0])) == 6
    d = (out[True][1] - out[False][1]).abs().max().item()
    print('[agent.train] max weight difference graph vs eager after 6 iterations: %.2e' % d)
    assert d < 3e-4 and torch.isfinite(out[True][1]).all()     # (Adam: a near-zero gradient whose sign differs moves a weight by lr per step)


def test_agent_test_replays_one_graph_per_minibatch(world):
    """Seq2SeqAgent.test (follower.py:987-999) on the device environment: every minibatch after the first is one replay of
    a captured inference rollout over a fixed-shape batch (test_graph), the next minibatch encoded under it -- the results
    of the loop that issues every rollout launch by launch (instructions padded to the minibatch's longest there)."""
    out = {}
    for graph in (False, True):
        ag, oe, od, weights = _fresh_agent(world, False)
        ag.test_graph = graph
        res = ag.test(use_dropout=False, feedback='argmax')
        again = ag.test(use_dropout=False, feedback='argmax')
        assert sorted(res) == sorted(again)
        assert (len(ag.__dict__.get('_test_graphs', {})) == 1) == graph
        out[graph] = (res, list(ag.losses), ag._engine.fallbacks)
    a, b = out[True][0], out[False][0]
    assert sorted(a) == sorted(b) and len(a) > 0
    for k in a:
        assert a[k]['instr_id'] == b[k]['instr_id'] and a[k]['actions'] == b[k]['actions']
        assert a[k]['trajectory'] == b[k]['trajectory']
        np.testing.assert_allclose(a[k]['scores'], b[k]['scores'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out[True][1], out[False][1], rtol=2e-5)
    assert out[True][2] == out[False][2] == 0


def test_a_replay_issued_ahead_is_not_used_once_the_weights_changed(world):
    """Seq2SeqAgent._rollout_on_graph issues minibatch k + 1's replay behind minibatch k's download; a caller that changes
    the weights between two rollouts (in place: the parameters' versions move) must get minibatch k + 1 under the NEW
    weights -- the rollout of the launch-by-launch path with the same changes."""
    out = {}
    for graph in (True, False):
        ag, oe, od, weights = _fresh_agent(world, False)
        ag.test_graph = graph
        ag.feedback = 'argmax'
        for m in (ag.encoder, ag.decoder):
            m.eval()
        with torch.no_grad():
            first = ag.rollout()
            assert ('_rollout_inflight' in ag.__dict__) == graph
            for p_ in ag.decoder.parameters():
                p_.mul_(0.7)
            second = ag.rollout()
            third = ag.rollout()                               # (and the pipeline goes on under the new weights)
        out[graph] = [[(t['instr_id'], t['actions'], t['scores']) for t in r] for r in (first, second, third)]
    for a, b in zip(out[True], out[False]):
        assert [x[:2] for x in a] == [x[:2] for x in b]
        for x, y in zip(a, b):
            np.testing.assert_allclose(x[2], y[2], rtol=0, atol=2e-5)


def test_preparing_the_next_minibatch_under_the_replay_changes_nothing(world):
    """agents.Seq2SeqAgent.prepare_ahead: minibatch i + 1 is drawn from the environment and encoded while replay i runs and
    reaches the device as one pinned copy (nav.DeviceNavBatch._pack_for_load) -- the same minibatches in the same order:
    losses and weights are bit-identical to the loop that prepares every minibatch after the previous loss was read."""
    out = {}
    for ahead in (False, True):
        ag, oe, od, weights = _fresh_agent(world, True, lr=1e-4)
        ag.prepare_ahead = ahead
        ag.train(oe, od, 4, feedback='teacher')
        first = list(ag.losses)
        ag.train(oe, od, 3, feedback='teacher')
        out[ahead] = (first + list(ag.losses), weights(), [it['instr_id'] for it in ag.env.batch])
    assert out[True][0] == out[False][0], (out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1])
    assert out[True][2] == out[False][2]                       # (the environment ends on the same minibatch)


def test_agent_train_on_graphs_survives_a_starved_persistent_launch(world):
    """A fault word raised inside a replayed iteration: the guarded optimizer steps do nothing on the device
    (sf_adam_step_dev), the agent sees the word where it reads the loss, restores its step counters and trains that
    minibatch again on the per-step kernels."""
    from speaker_follower_amd import runtime
    ag, oe, od, weights = _fresh_agent(world, True)
    ag.train(oe, od, 2, feedback='sample')
    tg = ag._train_graph_state[1]
    w0 = weights()
    with torch.cuda.stream(tg.stream):
        fw = runtime.fault_word(torch.device('cuda', 0))       # the capture stream's workspace: what the graph raises
    replay = tg.replay

    def poisoned():
        fw.fill_(runtime.FAULT_ENC_BWD)                            # as if the backward's persistent launch had starved
        st = replay()
        torch.cuda.synchronize()
        assert torch.equal(weights(), w0)                          # the guarded steps did not touch the weights
        return st
    tg.replay = poisoned
    try:
        ag.train(oe, od, 1, feedback='sample')
    finally:
        tg.replay = replay
    assert ag._engine.fallbacks == 1 and oe.host_steps() == [3] and od.host_steps() == [3]
    w1 = weights()
    assert torch.isfinite(w1).all() and not torch.equal(w1, w0) and np.isfinite(ag.losses).all()
    ag.train(oe, od, 2, feedback='sample')                         # and the graph goes on behind it
    assert oe.host_steps() == [5] and torch.isfinite(weights()).all()


def test_a_fault_in_the_middle_of_pipelined_replays_is_repaired_in_order(world):
    """Seq2SeqAgent._replay_pipelined queues replay i + 1 before it reads iteration i's fault words.  A fault raised
    inside replay i: neither i nor i + 1 stepped (the word stays raised on the device), the host re-issues minibatch i on
    the per-step kernels, queues minibatch i + 1 again, and the loop ends with every iteration trained once, on the
    minibatches the serial loop trains on, in its order."""
    from speaker_follower_amd import runtime
    seen = {}
    for pipelined in (True, False):
        ag, oe, od, weights = _fresh_agent(world, True)
        ag.pipeline_replays = pipelined
        ag.train(oe, od, 2, feedback='teacher')
        tg = ag._train_graph_state[1]
        with torch.cuda.stream(tg.stream):
            fw = runtime.fault_word(torch.device('cuda', 0))
        replay, calls, loaded = tg.replay, [], []
        load = ag._train_graph_state[2].load

        def counting_load(items, host=None):
            loaded.append([it['instr_id'] for it in items])
            return load(items, host)

        def poisoned():
            calls.append(len(calls))
            if len(calls) == 2:                                    # the second of this call's replays
                fw.fill_(runtime.FAULT_ENC_BWD)
            return replay()
        tg.replay = poisoned
        ag._train_graph_state[2].load = counting_load
        try:
            ag.train(oe, od, 5, feedback='teacher')
        finally:
            tg.replay = replay
            ag._train_graph_state[2].load = load
        assert ag._engine.fallbacks == 1 and oe.host_steps() == [7] and od.host_steps() == [7]
        assert len(ag.losses) == 5 and np.isfinite(ag.losses).all() and torch.isfinite(weights()).all()
        # every minibatch of the call, in order, once -- apart from the reloads of the repair
        order = []
        for ids in loaded:
            if ids not in order:
                order.append(ids)
        seen[pipelined] = (order, [it['instr_id'] for it in ag.env.batch], len(calls))
    assert seen[True][0] == seen[False][0] and len(seen[True][0]) == 5 and seen[True][1] == seen[False][1]
    assert seen[True][2] == 6 and seen[False][2] == 5              # (the replay queued behind the fault ran twice)


def test_agent_train_adopts_the_references_own_torch_adam(world):
    """train.py:263-268 hands Seq2SeqAgent.train two plain torch.optim.Adam objects.  The agent mirrors each by an
    optim.FusedAdam (same parameters, hyper-parameters and state, taken over at the start of a train() call and handed back
    at its end) and replays whole iterations; losses, weights and the optimizers' own state follow the loop in which torch's
    Adam steps on launch-by-launch iterations; a changed learning rate is a new graph."""
    out = {}
    for graph in (False, True):
        ag, oe, od, weights = _fresh_agent(world, graph, lr=1e-4, torch_adam=True)
        ag.adopt_torch_adam = graph
        ag.train(oe, od, 3, feedback='teacher')
        first = list(ag.losses)
        assert (ag.__dict__.get('_train_graph_state') is not None) == graph
        for o in (oe, od):
            steps = {int(st['step']) for st in o.state.values()}
            assert steps == {3} and len(o.state) == len(o.param_groups[0]['params'])
        for g in od.param_groups:
            g['lr'] = 5e-5                                          # (a schedule: the next call captures anew)
        ag.train(oe, od, 2, feedback='teacher')
        out[graph] = (first + list(ag.losses), weights(), {k: v.clone() for k, v in next(iter(od.state.values())).items()
                                                           if torch.is_tensor(v) and v.dim() > 0})
        assert {int(st['step']) for st in od.state.values()} == {5}
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-4)
    d = (out[True][1] - out[False][1]).abs().max().item()
    print('[agent.train, torch Adam adopted] max weight difference after 5 iterations: %.2e' % d)
    assert d < 3e-4 and torch.isfinite(out[True][1]).all()
    for k in out[True][2]:                                           # the moments torch's Adam would hold
        ref = out[False][2][k]
        assert float((out[True][2][k] - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-12, k
