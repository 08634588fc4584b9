"""GPU: the Seq2SeqAgent / Seq2SeqSpeaker mirrors (observation-dictionary interface of the
reference agents) against the already golden-checked fused engines on the same batches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speaker_follower_amd import synth                                # noqa: E402


@pytest.fixture(scope='module')
def setup():
    from speaker_follower_amd import model, features, follower, agents, env_synth
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights(101)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    fb = synth.follower_batch(seed=7, batch=8, steps=10, n_viewpoints=64, min_len=3, max_len=19, a_max=8)
    table = synth.feature_table(7, 64)
    env = env_synth.SyntheticR2REnv(fb, table)
    agent = agents.Seq2SeqAgent(env, '/tmp/sf_agent_results.json', enc, dec, episode_len=10)
    engine = follower.FollowerEngine(enc, dec, features.FeatureStore(table))
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    return agent, engine, batch, fb, enc, dec


def test_agent_teacher_rollout_matches_engine(setup):
    agent, engine, batch, fb, enc, dec = setup
    for m in (enc, dec):
        m.zero_grad(set_to_none=True)
    agent.feedback = 'teacher'
    traj = agent._rollout_with_loss()
    agent.loss.backward()
    g_agent = {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None}
    for m in (enc, dec):
        m.zero_grad(set_to_none=True)
    st = engine.rollout(batch, 10, 'teacher', train=False)
    st.loss.backward()
    np.testing.assert_allclose(float(agent.loss), float(st.loss), rtol=1e-5)
    acts = st.actions.cpu().numpy()
    for b, tr in enumerate(traj):
        n = len(tr['actions'])
        assert tr['actions'] == list(acts[:n, b])
        assert tr['actions'][-1] == 0 or n == 10
        np.testing.assert_allclose(tr['score'], st.step_scores[:n, b].sum().item(), rtol=1e-4, atol=1e-4)
    for k, p in dec.named_parameters():
        if k in g_agent and float(p.grad.abs().max()) > 1e-6:
            np.testing.assert_allclose(g_agent[k].cpu().numpy(), p.grad.cpu().numpy(), rtol=2e-4,
                                       atol=2e-6, err_msg=k)
    for m in (enc, dec):
        m.zero_grad(set_to_none=True)


def test_agent_test_loop_and_results(setup, tmp_path):
    agent, engine, batch, fb, enc, dec = setup
    agent.results_path = str(tmp_path / 'res.json')
    res = agent.test(use_dropout=False, feedback='argmax')
    assert len(res) == 8 and all('trajectory' in v for v in res.values())
    with torch.no_grad():
        st = engine.rollout(batch, 10, 'argmax', train=False)
    acts = st.actions.cpu().numpy()
    for b in range(8):
        tr = res['synth_%d' % b]
        assert tr['actions'] == list(acts[:len(tr['actions']), b])       # bit-exact argmax actions
    agent.write_results()
    agent.save(str(tmp_path / 'snap'))
    agent.load(str(tmp_path / 'snap'))
    with pytest.raises(RuntimeError, match='index-form'):         # dictionary env, no FeatureStore
        agent.beam_search(2)


def test_agent_sample_feedback_runs_and_respects_validity(setup):
    agent, engine, batch, fb, enc, dec = setup
    agent.feedback = 'sample'
    with torch.no_grad():
        traj = agent._rollout_with_loss()
    for b, tr in enumerate(traj):
        for t, a in enumerate(tr['actions']):
            assert 0 <= a < fb.a_num[t, b]
    with torch.no_grad():
        st = engine.rollout(batch, 10, 'sample', train=False)
    acts = st.actions.cpu().numpy()
    assert np.all(acts < fb.a_num[:10])


def test_agent_score_paths_matches_rollout_loss(setup):
    agent, engine, batch, fb, enc, dec = setup
    env = agent.env
    path_obs, path_actions, enc_instr = env.gold_obs_actions_and_instructions(10)
    with torch.no_grad():
        traj, loss = agent._score_obs_actions_and_instructions(path_obs, path_actions, enc_instr)
        st = engine.rollout(batch, 10, 'teacher', train=False)
    np.testing.assert_allclose(float(loss), float(st.loss), rtol=1e-5)
    for b, tr in enumerate(traj):
        assert tr['actions'] == path_actions[b]


def test_speaker_agent_matches_engine():
    from speaker_follower_amd import model, features, speaker, agents, env_synth
    d = synth.FULL
    senc_w, sdec_w = synth.speaker_weights(202)
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    fb = synth.follower_batch(seed=3, batch=6, steps=7, n_viewpoints=32, min_len=3, max_len=15, a_max=6)
    table = synth.feature_table(3, 32)
    env = env_synth.SyntheticR2REnv(fb, table)
    spk = agents.Seq2SeqSpeaker(env, '/tmp/sf_spk.json', enc, dec, instruction_len=20, max_episode_len=7)
    spk.feedback = 'teacher'
    out = spk.rollout()
    spk.loss.backward()
    assert len(out) == 6 and np.isfinite(float(spk.loss))
    for i, o in enumerate(out):
        want = list(fb.instr[i][:19]) + [2]
        assert o['word_indices'] == [int(w) for w in want[:len(o['word_indices'])]]
    assert float(dec.decoder2action.weight.grad.abs().sum()) > 0
    with torch.no_grad():
        res = spk.test(feedback='argmax')
    assert len(res) == 6
