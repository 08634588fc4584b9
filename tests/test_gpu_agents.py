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


def test_reference_style_training_loop_through_the_bare_name_shim(tmp_path):
    """What the reference's train.py does (make_env_and_models, train_val: train.py:176-205, 263-268, 92-125),
    written with its own statements over the compat modules: ImageFeatures-style feature holder ->
    R2RBatch(image_features_list, batch_size=, splits=, tokenizer=) -> Seq2SeqAgent(env, "", encoder, decoder,
    episode_len, max_instruction_length=) -> agent.train(Adam, Adam, n, feedback='sample') -> agent.env = val_env
    -> agent.test(...) -> agent.results.  The env built this way carries the feature store and no host table, so
    the agent walks it on the device by itself (one host sync per rollout)."""
    import json
    import os
    import sys
    from torch import optim
    from speaker_follower_amd import compat, features
    sys.path.insert(0, compat.path())
    try:
        for name in ('env', 'model', 'follower'):
            sys.modules.pop(name, None)
        from env import R2RBatch                                   # train.py:14
        from model import EncoderLSTM, AttnDecoderLSTM             # train.py:15
        from follower import Seq2SeqAgent                          # train.py:16
        import env as cenv
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        conn = os.path.join(root, 'tests', 'golden', 'connectivity')
        scans = ['gZ6f7yhEvPG', 'YmJkqBEsHnH', 'GdvgFV5R1Z5']
        graphs = {s: cenv.NavGraph(os.path.join(conn, s + '_connectivity.json')) for s in scans}
        rng = np.random.default_rng(12)
        for split, n in (('train', 24), ('val_seen', 12)):
            items = cenv.random_items(graphs, n, rng, min_len=4, max_len=12)
            data = [dict(path_id=1000 * (split == 'train') + i, scan=it['scan'], heading=it['heading'], path=it['path'],
                         distance=1.0, instructions=[' '.join('w%d' % t for t in it['instr_encoding'])])
                    for i, it in enumerate(items)]
            (tmp_path / ('R2R_%s.json' % split)).write_text(json.dumps(data))

        class Tok:                                                 # utils.Tokenizer.encode_sentence (utils.py:92-105)
            def encode_sentence(self, s):
                enc = np.array([int(w[1:]) for w in s.split()])
                return enc, len(enc)

        ids = [s + '_' + v for s in scans for v in graphs[s].ids]

        class Feats:                                               # MeanPooledImageFeatures: .store + get_name()
            store = features.FeatureStore(synth.feature_table(3, len(ids)), ids=ids)

            def get_name(self):
                return 'imagenet_mean_pooled'

        d = synth.FULL
        mk = lambda splits: R2RBatch([Feats()], batch_size=12, splits=splits, tokenizer=Tok(), nav_graph_path=conn,   # noqa: E731
                                     data_json=str(tmp_path / 'R2R_%s.json'))
        train_env, val_env = mk(['train']), mk(['val_seen'])
        enc_w, dec_w = synth.follower_weights_peaky(5)
        encoder = EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight']).cuda()
        decoder = AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat).cuda()
        agent = Seq2SeqAgent(train_env, "", encoder, decoder, 6, max_instruction_length=80)
        filt = lambda ps: [p for p in ps if p.requires_grad]       # noqa: E731  train.py:64-65
        eo = optim.Adam(filt(agent.encoder.parameters()), lr=1e-4, weight_decay=5e-4)
        do = optim.Adam(filt(agent.decoder.parameters()), lr=1e-4, weight_decay=5e-4)
        agent.train(eo, do, 3, feedback='sample')                  # train.py:100-101
        assert len(agent.losses) == 3 and all(np.isfinite(agent.losses))
        assert agent.nav_table is None and agent._engine is not None      # walked on the device, by itself
        agent.env = val_env                                        # train.py:111
        agent.test(use_dropout=True, feedback='sample', allow_cheat=True)
        assert len(agent.losses) >= 1
        agent.results_path = str(tmp_path / 'val_seen.json')
        agent.test(use_dropout=False, feedback='argmax')           # train.py:123
        agent.write_results()
        res = json.load(open(agent.results_path))
        assert sorted(res) == sorted(it['instr_id'] for it in val_env.data)
        for r in res.values():                                     # trajectories stay on the graph
            g = graphs[next(it['scan'] for it in val_env.data if it['instr_id'] == r['instr_id'])]
            vps = [p[0] for p in r['trajectory']]
            assert all(b == a or b in g.adj[a] for a, b in zip(vps, vps[1:]))
        base = str(tmp_path / 'snap')
        agent.save(base)                                           # follower.py:1022-1035
        assert all(os.path.exists(p) for p in agent._encoder_and_decoder_paths(base))
    finally:
        sys.path.remove(compat.path())
        for name in ('env', 'model', 'follower', 'speaker'):
            sys.modules.pop(name, None)


@pytest.mark.parametrize('feedback', ['teacher', 'argmax'])
def test_speaker_agent_index_form_path_equals_the_dense_module_path(feedback):
    """Seq2SeqSpeaker over an env WITHOUT a host feature table (index-form observations, what the bare-name shim
    builds): the fused engine path gives the words, scores and loss of the observation-dictionary path that
    mirrors speaker.py:123-202 on dense rows, and trains (train_speaker.py's loop)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import search_world as W
    from torch import optim
    from speaker_follower_amd import model, agents, features
    d = synth.FULL
    senc_w, sdec_w = synth.speaker_weights_peaky(W.SPEAKER_SEED)

    def modules():
        enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
        dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
        enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
        dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
        return enc.cuda().eval(), dec.cuda().eval()

    out = {}
    for dense in (True, False):
        env, table = W.build_world(dense=dense)
        enc, dec = modules()
        spk = agents.Seq2SeqSpeaker(env, '/tmp/sf_spk_%d.json' % dense, enc, dec, W.INSTRUCTION_LEN, max_episode_len=W.EPISODE_LEN)
        if not dense:
            spk.store = features.FeatureStore(table)
        env.reset_epoch()
        spk.feedback = feedback
        with torch.no_grad():
            res = spk.rollout()
        out[dense] = (res, float(spk.loss))
    a, b = out[True], out[False]
    assert [r['instr_id'] for r in a[0]] == [r['instr_id'] for r in b[0]]
    for ra, rb in zip(a[0], b[0]):
        assert ra['word_indices'] == rb['word_indices']
        np.testing.assert_allclose(rb['scores'], ra['scores'], rtol=2e-4, atol=2e-4)
        assert abs(ra['score'] - rb['score']) <= 3e-4 * max(1.0, abs(ra['score']))
        assert ra['words'] == rb['words']
    if feedback == 'teacher':                     # every step has live targets until the longest instruction ends
        np.testing.assert_allclose(b[1], a[1], rtol=1e-4)
    # train_speaker.py's loop on the index-form env (speaker.py:376-395)
    env, table = W.build_world(dense=False)
    enc, dec = modules()
    spk = agents.Seq2SeqSpeaker(env, '/tmp/sf_spk_train.json', enc, dec, W.INSTRUCTION_LEN, max_episode_len=W.EPISODE_LEN)
    spk.store = features.FeatureStore(table)
    filt = lambda ps: [p for p in ps if p.requires_grad]       # noqa: E731
    eo = optim.Adam(filt(enc.parameters()), lr=1e-3)
    do = optim.Adam(filt(dec.parameters()), lr=1e-3)
    env.reset_epoch()
    spk.train(eo, do, 1, feedback='teacher')
    first = spk.losses[0]
    for _ in range(6):
        env.reset_epoch()
        spk.train(eo, do, 1, feedback='teacher')
    assert np.isfinite(spk.losses[0]) and spk.losses[0] < first          # the same minibatch: the loss goes down


def test_speaker_agent_gold_routes_from_the_tables_equal_the_environment_walk():
    """Seq2SeqSpeaker.rollout (speaker.py:348-360) over an index-form environment: the minibatch's gold routes from the
    navigation tables (index_gold_routes, nav.NavTable.gold_routes) give the outputs and the loss of the lock-step walk
    of the host environment -- and train() runs on them."""
    import os
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import search_world as W
    from speaker_follower_amd import agents, features, model, optim
    rng_state = random.getstate()       # (the env reshuffles with the global generator when an epoch wraps, env.py:601-614:
    try:                                # tests that run later pin their minibatch order on its state)
        _speaker_gold_routes_body(W, agents, features, model, optim)
    finally:
        random.setstate(rng_state)


def _speaker_gold_routes_body(W, agents, features, model, optim):
    env, table = W.build_world(dense=False, n_items=24, batch=12, item_seed=5)
    d = synth.FULL
    w_enc, w_dec = synth.speaker_weights(W.SPEAKER_SEED)
    senc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    sdec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=w_dec['embedding.weight'])
    senc.load_state_dict({k: torch.tensor(v) for k, v in w_enc.items()})
    sdec.load_state_dict({k: torch.tensor(v) for k, v in w_dec.items()})
    senc.cuda().eval()
    sdec.cuda().eval()
    spk = agents.Seq2SeqSpeaker(env, '/tmp/sf_spk_gold.json', senc, sdec, W.INSTRUCTION_LEN, max_episode_len=W.EPISODE_LEN)
    spk.store = features.FeatureStore(table)
    out = {}
    for index in (True, False):
        spk.index_gold_routes = index
        env.reset_epoch()
        res = []
        with torch.no_grad():
            for fb in ('teacher', 'argmax'):
                spk.feedback = fb
                o = spk.rollout()
                res.append(([(x['instr_id'], x['word_indices'], x['score']) for x in o], float(spk.loss)))
        out[index] = res
    for (a, la), (b, lb) in zip(out[True], out[False]):
        assert [x[:2] for x in a] == [x[:2] for x in b]
        np.testing.assert_allclose([x[2] for x in a], [x[2] for x in b], rtol=0, atol=1e-5)
        assert abs(la - lb) <= 1e-6 * abs(lb)
    spk.index_gold_routes = True
    oe = optim.FusedAdam([p for p in senc.parameters() if p.requires_grad], lr=1e-4)
    od = optim.FusedAdam([p for p in sdec.parameters() if p.requires_grad], lr=1e-4)
    env.reset_epoch()
    spk.train(oe, od, 3, feedback='teacher')
    assert len(spk.losses) == 3 and np.isfinite(spk.losses).all()


def test_speaker_train_without_outputs_is_the_same_training():
    """Seq2SeqSpeaker.train: the iteration that issues forward and backward back to back, forms the next minibatch's
    routes under them and syncs once (train_without_outputs) against rollout() + loss.backward() (speaker.py:376-395):
    the same minibatches, kernels and dropout sites -- losses and weights bit-identical."""
    import os
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import search_world as W
    from speaker_follower_amd import agents, features, model, optim
    rng_state = random.getstate()
    try:
        env, table = W.build_world(dense=False, n_items=60, batch=12, item_seed=7)
        store = features.FeatureStore(table)
        d = synth.FULL
        w_enc, w_dec = synth.speaker_weights(W.SPEAKER_SEED)
        out = {}
        for fast in (False, True):
            senc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
            sdec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=w_dec['embedding.weight'])
            senc.load_state_dict({k: torch.tensor(v) for k, v in w_enc.items()})
            sdec.load_state_dict({k: torch.tensor(v) for k, v in w_dec.items()})
            senc.cuda()
            sdec.cuda()
            torch.manual_seed(3)
            spk = agents.Seq2SeqSpeaker(env, '/tmp/sf_spk_fast.json', senc, sdec, W.INSTRUCTION_LEN,
                                        max_episode_len=W.EPISODE_LEN)
            spk.store = store
            spk.train_without_outputs = fast
            oe = optim.FusedAdam([p for p in senc.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
            od = optim.FusedAdam([p for p in sdec.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
            env.reset_epoch()
            spk.train(oe, od, 4, feedback='teacher')             # (5 minibatches per epoch: no wrap, no reshuffle)
            w = torch.cat([p.detach().reshape(-1) for m in (senc, sdec) for p in m.parameters()]).clone()
            out[fast] = (list(spk.losses), w, [it['instr_id'] for it in env.batch], spk._engine.fallbacks)
        print('[speaker.train] losses', out[True][0])
        assert out[True][0] == out[False][0] and len(set(out[True][0])) == 4
        assert torch.equal(out[True][1], out[False][1])
        assert out[True][2] == out[False][2] and out[True][3] == out[False][3] == 0
    finally:
        random.setstate(rng_state)


def test_speaker_test_as_a_sweep_equals_the_loop_over_rollouts():
    """Seq2SeqSpeaker.test (speaker.py:397-414) decoded as one speaker.SpeakerSweep over the epoch's minibatches against the
    loop over rollout(): the same instructions, word for word, scores and losses to rounding -- before and AFTER training
    steps (the sweep's graphs are captured once and must follow the weights)."""
    import os
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import search_world as W
    from speaker_follower_amd import agents, features, model, optim
    rng_state = random.getstate()
    try:
        env, table = W.build_world(dense=False, n_items=60, batch=12, item_seed=9)
        d = synth.FULL
        w_enc, w_dec = synth.speaker_weights(W.SPEAKER_SEED)
        senc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
        sdec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=w_dec['embedding.weight'])
        senc.load_state_dict({k: torch.tensor(v) for k, v in w_enc.items()})
        sdec.load_state_dict({k: torch.tensor(v) for k, v in w_dec.items()})
        senc.cuda()
        sdec.cuda()
        spk = agents.Seq2SeqSpeaker(env, '/tmp/sf_spk_sweep.json', senc, sdec, W.INSTRUCTION_LEN, max_episode_len=W.EPISODE_LEN)
        spk.store = features.FeatureStore(table)
        oe = optim.FusedAdam([p for p in senc.parameters() if p.requires_grad], lr=1e-3)
        od = optim.FusedAdam([p for p in sdec.parameters() if p.requires_grad], lr=1e-3)

        def both_ways():
            out = []
            for after in (10 ** 9, 0):
                spk.sweep_test_after = after
                random.seed(17)                        # (the epoch wraps inside test(): the same reshuffle both ways)
                env.data.sort(key=lambda it: it['instr_id'])
                res = spk.test(use_dropout=False, feedback='argmax')
                out.append(({k: (v['word_indices'], v['score'], v['scores']) for k, v in res.items()}, list(spk.losses),
                            [it['instr_id'] for it in env.batch]))
            loop, sweep = out
            assert ('_test_sweep' in spk.__dict__)
            assert sorted(loop[0]) == sorted(sweep[0]) and len(loop[0]) == 60 and loop[2] == sweep[2]
            for k in loop[0]:
                assert loop[0][k][0] == sweep[0][k][0], k
                np.testing.assert_allclose(sweep[0][k][2], loop[0][k][2], rtol=0, atol=2e-5)
                assert abs(loop[0][k][1] - sweep[0][k][1]) < 1e-4
            np.testing.assert_allclose(sweep[1], loop[1], rtol=1e-5)
            return loop[0]
        first = both_ways()
        env.reset_epoch()
        spk.train(oe, od, 5, feedback='teacher')
        second = both_ways()
        assert any(first[k][0] != second[k][0] for k in first)          # (the training steps changed what is generated)
        assert spk._test_sweep[1].fallbacks == 0
    finally:
        random.setstate(rng_state)
