"""GPU, FULL-SIZE configs[1]: the 10 567-viewpoint feature table (36 x 2048 fp32 per viewpoint = 3.1 GB: rows
>= 7 282 start beyond 2^31 BYTES, rows >= 3 641 beyond 2^31 / 2 ... i.e. every 32-bit offset trap is on the
path) and the environment of all 90 connectivity graphs.

* a1 / a4 / a7 / a11 with viewpoints forced to rows {0, 7 281, 7 282, 10 566}: indexed gather == dense ==
  oracle (env.py:380-383 lookup feeding model.py:310-326, 342-352, follower.py:476-505), follower rollout
  and speaker scoring through the engines;
* device-resident environment: >= 200 sampled items from >= 10 scans, trajectories / actions / loss equal to
  the per-step host loop (agents._rollout_with_loss, the mirror of follower.py:430-539) for argmax and
  sample feedback.
"""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import np_env, np_model                                   # noqa: E402  (checker only)
from tests.tol import assert_logits_close                            # noqa: E402
from speaker_follower_amd import synth                                # noqa: E402

TOL = dict(rtol=1e-4, atol=1e-4)
N_VP = 10567
EDGE_ROWS = np.array([0, 7281, 7282, 10566], np.int32)      # 7282 * 36 * 2048 * 4 B = 2^31 + 524 288


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


@pytest.fixture(scope='module')
def big():
    """The full-size table on the device (generated there, like bench.py) + a FeatureStore over it."""
    import bench
    from speaker_follower_amd import features
    table = bench.device_table(N_VP, 1234, torch.device('cuda', 0))
    assert table.numel() * 4 > (1 << 31)                       # 3.1 GB: past 2^31 bytes
    store = features.FeatureStore(table, device='cuda')
    return table, store


def _small(table, rows):
    """Host copy of `rows` of the device table: the oracle's table, indexed 0 .. len(rows)-1."""
    return table[torch.from_numpy(np.asarray(rows, np.int64)).cuda()].cpu().numpy()


def test_table_rows_beyond_two_gib_are_addressed_correctly(big):
    """a11 gathers at the edge rows: bit-exact against host arithmetic on the same rows."""
    table, store = big
    B = 8
    rng = np.random.default_rng(5)
    vp_small = np.arange(B, dtype=np.int32) % len(EDGE_ROWS)
    vp = EDGE_ROWS[vp_small]
    view = rng.integers(0, 36, B).astype(np.int32)
    small = _small(table, EDGE_ROWS)
    loc = np_env.static_loc_embeddings()
    X = np.stack([np_env.panorama_feature(small[vp_small[b]], view[b], loc) for b in range(B)])
    vp_d, view_d = dev(vp), dev(view)
    np.testing.assert_array_equal(store.gather_panorama(vp_d, view_d).cpu().numpy(), X)
    assert float(np.abs(small[1] - small[2]).max()) > 0.1      # neighbouring rows differ: an off-by-one row would show


def test_visual_attention_and_scoring_at_the_edge_rows(big):
    """a1 and a4: indexed == dense == oracle with every sample on an edge row."""
    from speaker_follower_amd import ops, features
    table, store = big
    d = synth.FULL
    H, F, D, V = d.hidden, d.feat, d.dot, d.views
    B, A = 8, 9
    rng = np.random.default_rng(17)
    small = _small(table, EDGE_ROWS)
    loc = np_env.static_loc_embeddings()
    rnd = lambda *s, scale=1.0: (rng.standard_normal(s) * scale).astype(np.float32)      # noqa: E731
    # ---- a1
    vp_small = np.arange(B, dtype=np.int32) % len(EDGE_ROWS)
    vp, view = EDGE_ROWS[vp_small], rng.integers(0, 36, B).astype(np.int32)
    X = np.stack([np_env.panorama_feature(small[vp_small[b]], view[b], loc) for b in range(B)])
    w = [rnd(D, H, scale=H ** -0.5), rnd(D, scale=0.1), rnd(D, F, scale=4 * F ** -0.5), rnd(D, scale=0.1)]
    h = rnd(B, H)
    ref_out, ref_alpha = np_model.visual_soft_dot_attention(h, X, *w)
    assert float(ref_alpha.max()) > 0.2                        # a peaked distribution, not 1/36 everywhere
    wd = [dev(a) for a in w]
    vp_d, view_d = dev(vp), dev(view)
    Xg = store.gather_panorama(vp_d, view_d)
    for pano in (ops.pano_dense(Xg), store.pano(vp_d, view_d)):
        out, alpha, _, _ = ops.visual_attention_fwd(wd, pano, B, V, F, dev(h))
        np.testing.assert_allclose(out.cpu().numpy(), ref_out, **TOL)
        np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, **TOL)
    # ---- a4 (+ the candidate gather of a11)
    fb = synth.follower_batch(seed=3, batch=B, steps=1, n_viewpoints=len(EDGE_ROWS), a_max=A)
    _, all_u, is_valid = np_env.dense_follower_step(small, loc, fb, 0)
    pad = np.zeros((B, A, F), np.float32)
    pad[:, :all_u.shape[1]] = all_u
    w4 = [rnd(D, H, scale=H ** -0.5), rnd(D, scale=0.1), rnd(D, F, scale=F ** -0.5), rnd(D, scale=0.1),
          rnd(1, D, scale=D ** -0.5), rnd(1, scale=0.1)]
    ref = np_model.eltwise_prod_scoring(h, pad, *w4)
    wd4 = [dev(a) for a in w4]
    idx = [dev(EDGE_ROWS[fb.vp[0]]), dev(fb.cand_view[0]),
           dev(features.cand_sincos(fb.cand_heading[0], fb.cand_elevation[0])), dev(fb.a_num[0])]
    Ug, validg = store.gather_candidates(*idx)
    np.testing.assert_array_equal(Ug.cpu().numpy(), pad)
    np.testing.assert_array_equal(validg.cpu().numpy()[:, :is_valid.shape[1]], is_valid)
    for cnd in (ops.cands_dense(Ug), store.cands(*idx, A)):
        logit, _, _, _ = ops.eltwise_prod_scoring_fwd(wd4, cnd, B, A, F, dev(h))
        np.testing.assert_allclose(logit.cpu().numpy(), ref, **TOL)


def _follower(seed=303):
    from speaker_follower_amd import model
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return enc.cuda().eval(), dec.cuda().eval(), enc_w, dec_w


def test_follower_rollout_on_edge_rows_matches_oracle(big):
    """a6 + a7 through the engine (pipelined paired schedule, in-kernel gathers) with every observation on an
    edge row: logits 1e-4, actions bit-exact, loss."""
    from speaker_follower_amd import follower
    table, store = big
    enc, dec, enc_w, dec_w = _follower()
    B, S = 12, 4
    fb = synth.follower_batch(seed=11, batch=B, steps=S, n_viewpoints=len(EDGE_ROWS), min_len=5, max_len=30)
    fb_big = copy.copy(fb)
    fb_big.vp = EDGE_ROWS[fb.vp]
    small = _small(table, EDGE_ROWS)
    eng = follower.FollowerEngine(enc, dec, store)
    with torch.no_grad():
        st = eng.rollout(follower.DeviceFollowerBatch.from_synth(fb_big), S, 'argmax', train=False)
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, 80, reverse=True)
    loc = np_env.static_loc_embeddings()
    ref = np_model.follower_rollout(enc_w, dec_w, seq, lens, mask, S,
                                    lambda t: np_env.dense_follower_step(small, loc, fb, t),
                                    fb.target, 'argmax', synth.FULL.feat, early_exit=False)
    n = len(ref['logits'])
    assert n == S
    np.testing.assert_array_equal(st.actions.cpu().numpy()[:n], ref['actions'])
    lg = st.logits.cpu().numpy()
    for t in range(n):
        a = ref['logits'][t].shape[1]
        assert_logits_close(lg[t][:, :a], ref['logits'][t], 'full-table follower rollout, step %d' % t)
    np.testing.assert_allclose(float(st.loss), float(ref['loss']), rtol=1e-4, atol=1e-5)


def test_speaker_scoring_on_edge_rows_matches_oracle(big):
    """a8 - a10 through SpeakerEngine (visual attention per path step + action gather from the table)."""
    from speaker_follower_amd import model, speaker
    table, store = big
    d = synth.FULL
    senc_w, sdec_w = synth.speaker_weights_peaky(202)
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    sb = synth.speaker_batch(seed=9, batch=6, n_viewpoints=len(EDGE_ROWS), min_len=3, max_len=25)
    sb_big = copy.copy(sb)
    sb_big.vp = EDGE_ROWS[sb.vp]
    small = _small(table, EDGE_ROWS)
    loc = np_env.static_loc_embeddings()
    acts, feats, pmask = np_env.dense_speaker_inputs(sb, small, loc)
    seq, _, _ = np_env.batch_instructions_from_encoded(sb.instr, 80, reverse=False)
    S = 12
    ref = np_model.speaker_score(senc_w, sdec_w, acts, feats, pmask, seq, S, 'teacher')
    with torch.no_grad():
        st = speaker.SpeakerEngine(enc, dec, store).score(speaker.DeviceSpeakerBatch.from_synth(sb_big), S, 'teacher',
                                                          train=False)
    n = len(ref['logits'])
    np.testing.assert_allclose(st.ctx.cpu().numpy(), ref['ctx'], **TOL)
    lg = st.logits.cpu().numpy()
    for t in range(n):
        assert_logits_close(lg[t], ref['logits'][t], 'full-table speaker scoring, word step %d' % t)
    np.testing.assert_allclose(float(st.loss), float(ref['loss']), rtol=1e-4)
    np.testing.assert_array_equal(np.argmax(lg[:n], axis=2), np.stack([np.argmax(l, axis=1) for l in ref['logits']]))


class _DeviceRows:
    """host_table stand-in for R2RIndexEnv: row r -> numpy [36, 2048], fetched from the device table on first use."""

    def __init__(self, table):
        self.table, self.cache = table, {}

    def __getitem__(self, r):
        r = int(r)
        if r not in self.cache:
            self.cache[r] = self.table[r].cpu().numpy()
        return self.cache[r]


@pytest.mark.parametrize('feedback', ['argmax', 'sample'])
def test_full_environment_device_rollouts_equal_the_host_loop(big, feedback):
    """All 90 graphs tabulated on the device; 2 minibatches x 100 items (>= 10 scans each): trajectories,
    actions, per-step scores and the loss of the one-sync device rollout equal the reference-style loop that
    syncs the actions to the host and steps / observes the environment in Python every step."""
    from speaker_follower_amd import agents, bench_extras, follower, nav
    table, store = big
    enc, dec, _, _ = _follower()
    EP = 6
    total, scans_seen = 0, set()
    for seed in (21, 22):
        e, nt = bench_extras.full_world(store, batch=100, seed=seed)
        assert nt.n_rows == N_VP and len(nt.scans) == 90 and nt.A == 14
        e.host_table = _DeviceRows(table)
        agent = agents.Seq2SeqAgent(e, '/tmp/sf_full.json', enc, dec, episode_len=EP)
        agent.store = store
        e.reset_epoch()
        agent.feedback = feedback
        agent._sample_count = 0
        with torch.no_grad():
            want = agent._rollout_with_loss()
        want_loss = float(agent.loss)
        items = list(e.batch)
        eng = follower.FollowerEngine(enc, dec, store)
        eng.dropout_seed = agent._sample_seed ^ 0x1B873593
        eng.site_next = 1
        navb = nav.DeviceNavBatch(nt, items, EP)
        with torch.no_grad():
            st = eng.rollout(navb, EP, feedback, train=False)
        got = navb.trajectories(st)
        assert [g['instr_id'] for g in got] == [w['instr_id'] for w in want]
        moved = 0
        for g, w in zip(got, want):
            assert g['actions'] == [int(a) for a in w['actions']], (g['instr_id'], feedback)
            assert [p[0] for p in g['trajectory']] == [p[0] for p in w['trajectory']]
            for pg, pw in zip(g['trajectory'], w['trajectory']):
                assert pg[1] == pytest.approx(pw[1], abs=1e-12) and pg[2] == pytest.approx(pw[2], abs=1e-12)
            np.testing.assert_allclose(g['scores'], w['scores'], rtol=2e-4, atol=2e-4)
            moved += len({p[0] for p in g['trajectory']}) > 1
        assert moved >= len(got) // 2
        np.testing.assert_allclose(float(st.loss), want_loss, rtol=1e-4)
        total += len(got)
        scans_seen |= {it['scan'] for it in items}
        assert any(e.row_of[it['scan'] + '_' + it['path'][0]] >= 7282 for it in items)    # starts beyond the 2 GiB mark
    assert total >= 200 and len(scans_seen) >= 10
