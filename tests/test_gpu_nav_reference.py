"""GPU: the device-resident environment (nav.NavTable on the device, csrc/sf_nav.hip / the env step fused into the scoring
launch) against the REFERENCE's own env.py on the real R2R_sub_val_seen split (tests/golden/g15_env_reference.json.gz,
written by tests/golden/make_golden_env.py from tasks/R2R/env.py:126-224, 742-854): a teacher-forced device rollout
must visit the viewpoints and views, choose the teacher actions and see the candidate counts that the reference's
`gold_obs_actions_and_instructions` recorded, for every instruction of the split."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import r2r_val_seen as VS                                                   # noqa: E402


@pytest.fixture(scope='module')
def world():
    from speaker_follower_amd import model, features, synth, nav
    items, _ = VS.load_items()
    gold = VS.load_env_golden()
    env, row_of, n = VS.build_env(items, batch_size=gold['config']['batch'])
    table = torch.empty(n, 36, 2048, device='cuda').normal_(0.0, 0.5).clamp_(min=0.0)    # (values do not matter here)
    store = features.FeatureStore(table)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(303)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return items, gold, env, store, nav.NavTable(env, store), enc.cuda().eval(), dec.cuda().eval()


def test_device_tables_hold_the_reference_sweep(world):
    """The DEVICE copies of the candidate tables at the fixture's states: next viewpoint, absViewIndex, sin/cos of the
    relative angles (fp32 of the reference's float64 angles)."""
    items, gold, env, store, nt, enc, dec = world
    a_num, nxt, cv, sc = (t.cpu().numpy() for t in (nt.a_num, nt.next_row, nt.cand_view, nt.sincos))
    for s in gold['states']:
        sid = nt.row_of[(s['scan'], s['viewpoint'])] * 36 + s['viewIndex']
        assert a_num[sid] == len(s['adj'])
        for a, (vp2, view2, rh, re) in enumerate(s['adj']):
            if a == 0:
                assert nxt[sid, 0] == sid // 36
                continue
            assert nt.vp_of[nxt[sid, a]] == (s['scan'], vp2) and cv[sid, a] == view2
            want = np.array([np.sin(rh), np.cos(rh), np.sin(re), np.cos(re)], np.float32)
            np.testing.assert_array_equal(sc[sid, a], want)


@pytest.mark.parametrize('fused', [True, False])
def test_teacher_forced_device_rollout_walks_the_reference_routes(world, fused):
    from speaker_follower_amd import follower, nav
    items, gold, _, store, nt, enc, dec = world
    env, _, _ = VS.build_env(items, batch_size=gold['config']['batch'])
    S = gold['config']['max_steps']
    eng = follower.FollowerEngine(enc, dec, store)
    eng.fused_env_step = fused
    seen = set()
    for _ in range(9):
        env._next_minibatch(True)
        batch = list(env.batch)
        navb = nav.DeviceNavBatch(nt, batch, S)
        with torch.no_grad():
            st = eng.rollout(navb, S, 'teacher', train=False)
        rows, views = navb.row[:S + 1].cpu().numpy(), navb.view[:S + 1].cpu().numpy()
        a_num, target = navb.a_num[:S + 1].cpu().numpy(), navb.target[:S + 1].cpu().numpy()
        acts = st.actions.cpu().numpy()
        for b, it in enumerate(batch):
            w = gold['routes'][it['instr_id']]
            n = len(w['actions'])
            assert [nt.vp_of[r][1] for r in rows[:n + 1, b]] == w['viewpoints']
            assert views[:n + 1, b].tolist() == w['views']
            assert target[:n, b].tolist() == w['actions'] and acts[:n, b].tolist() == w['actions']
            assert a_num[:n + 1, b].tolist() == w['a_num']
            assert (target[n:S, b] == -1).all()                            # ended: ignored by the loss (follower.py:376-381)
            seen.add(it['instr_id'])
        # the trajectories in the reference's result format: snapped headings, the stop's duplicated final pose
        for res in navb.trajectories(st):
            w = gold['routes'][res['instr_id']]
            assert [p[0] for p in res['trajectory']] == w['viewpoints']
            np.testing.assert_allclose([p[1] for p in res['trajectory']], w['headings'], atol=1e-12)
    assert len(seen) == 782
