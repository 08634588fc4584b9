"""GPU, EXPERIMENT (skipped unless libsf_experimental.so was built: `python -m speaker_follower_amd.build
--experimental`): the follower's decode loop as ONE persistent launch (csrc/experimental/sf_mega.hip,
sf_follower_decode_persistent) against the per-stage engine, which is itself pinned to the reference's goldens
(tests/test_gpu_follower.py, test_gpu_hard_parity.py).  The comparison is HIP path vs HIP path: it carries no
parity credit of its own, it keeps a frozen, slower-than-default experiment honest."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speaker_follower_amd import _lib                                 # noqa: E402
if not os.path.exists(_lib.EXP_LIB_PATH):
    pytest.skip('libsf_experimental.so not built (frozen experiment, outside the product library)',
                allow_module_level=True)

from speaker_follower_amd import synth                                # noqa: E402


def make_engine(peaky=True):
    from speaker_follower_amd import model, features, follower
    d = synth.FULL
    enc_w, dec_w = (synth.follower_weights_peaky if peaky else synth.follower_weights)(303)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    store = features.FeatureStore(synth.feature_table(8, 256))
    return follower.FollowerEngine(enc, dec, store)


def rollouts(B, S, feedback='argmax', seed=47, debug=True):
    from speaker_follower_amd import follower
    eng = make_engine()
    fb = synth.follower_batch(seed=seed, batch=B, steps=S, n_viewpoints=256)
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    with torch.no_grad():
        eng.fold_inference = True                  # same folded Linears as the persistent kernel
        ref = eng.rollout(batch, S, feedback, train=False)
        torch.cuda.synchronize()
        eng.persistent_decode, eng.persistent_debug_tapes = True, debug
        eng.site_next, eng.iteration = 0, 0        # the same sampling sites as the first rollout
        got = eng.rollout(batch, S, feedback, train=False)
        torch.cuda.synchronize()
    assert got.persistent, 'the persistent launch was refused for this shape'
    return eng, batch, ref, got


def compare(ref, got, S, B, tapes=True):
    F = 2176
    if tapes:
        # step by step, so that the first deviating quantity is the one reported
        for t in range(S):
            for name, a, b in (
                    ('feat', got.tape['xin'][t][:, F:], ref.tape['xin'][t][:, F:]),
                    ('u', got.tape['xin'][t][:, :F], ref.tape['xin'][t][:, :F]),
                    ('h1', got.tape['h1'][t], ref.tape['h1'][t]),
                    ('c1', got.tape['c1'][t], ref.tape['c1'][t]),
                    ('t_text', got.tape['t_text'][t], ref.tape['t_text'][t]),
                    ('wc', got.tape['cat2'][t][:, :512], ref.tape['cat2'][t][:, :512]),
                    ('h_tilde', got.tape['h_tilde'][t], ref.tape['h_tilde'][t]),
                    ('logit', got.logits[t], ref.logits[t])):
                assert not torch.isnan(a).any(), 'step %d %s has NaN' % (t, name)
                scale = max(float(b[torch.isfinite(b)].abs().max()), 1e-3)
                fin = torch.isfinite(b)
                assert (torch.isfinite(a) == fin).all(), 'step %d %s: mask pattern differs' % (t, name)
                err = float((a[fin] - b[fin]).abs().max())
                assert err <= 2e-5 * scale + 1e-6, 'step %d %s: max err %.3e (scale %.3e)' % (t, name, err, scale)
            assert torch.equal(got.actions[t], ref.actions[t]), 'step %d actions differ' % t
    assert torch.equal(got.actions, ref.actions)
    assert torch.equal(got.target_used, ref.target_used)
    assert torch.equal(got.ended, ref.ended)
    torch.testing.assert_close(got.step_scores, ref.step_scores, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got.ce_term, ref.ce_term, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got.live, ref.live, rtol=0, atol=0)
    torch.testing.assert_close(got.h, ref.h, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got.c, ref.c, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got.loss, ref.loss, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('B,S', [(100, 20), (128, 5), (16, 6), (37, 9), (3, 4), (1, 3)])
def test_persistent_decode_matches_the_per_stage_engine(B, S):
    eng, batch, ref, got = rollouts(B, S)
    compare(ref, got, S, B)


@pytest.mark.parametrize('feedback', ['teacher', 'sample'])
def test_persistent_decode_feedback_modes(feedback):
    eng, batch, ref, got = rollouts(40, 8, feedback)
    compare(ref, got, 8, 40, tapes=False)


def test_persistent_decode_falls_back_outside_its_envelope():
    """B > 128: sf_follower_decode_persistent answers SF_ERR_UNSUPPORTED and the engine runs the per-stage
    episode instead (same results as an engine that never asked)."""
    from speaker_follower_amd import follower
    eng = make_engine()
    fb = synth.follower_batch(seed=5, batch=130, steps=3, n_viewpoints=256)
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    with torch.no_grad():
        ref = eng.rollout(batch, 3, 'argmax', train=False)
        eng.persistent_decode = True
        eng.site_next, eng.iteration = 0, 0
        got = eng.rollout(batch, 3, 'argmax', train=False)
        torch.cuda.synchronize()
    assert got.persistent is False
    assert torch.equal(got.actions, ref.actions) and torch.equal(got.logits, ref.logits)


def test_persistent_decode_is_repeatable_and_leaves_no_state():
    eng, batch, ref, got = rollouts(100, 12, debug=False)
    with torch.no_grad():
        again = eng.rollout(batch, 12, 'argmax', train=False)
        torch.cuda.synchronize()
    assert torch.equal(again.actions, got.actions)
    assert torch.equal(again.logits, got.logits)
    assert torch.equal(again.h, got.h)
    compare(ref, again, 12, 100, tapes=False)


def test_persistent_decode_timing():
    from speaker_follower_amd import _lib
    B, S = 100, 20
    eng, batch, ref, got = rollouts(B, S, debug=False)
    run = lambda: eng.rollout(batch, S, 'argmax', train=False)       # noqa: E731
    with torch.no_grad():
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        with _lib.kernel_profile(_lib.experimental()) as prof:
            for _ in range(5):
                run()
        rows = sorted(prof.rows.items(), key=lambda kv: -kv[1]['total_us'])
        for k, v in rows[:6]:
            print('    %-50s calls %3d avg %9.1f us' % (k[:50], v['calls'] // 5, v['avg_us']))
        mk = [v for k, v in prof.rows.items() if 'mega_kernel' in k][0]
        print('decode loop, %d steps in one launch: %.1f us = %.2f us per step' % (S, mk['avg_us'], mk['avg_us'] / S))
        trace = torch.zeros(256 * 32, dtype=torch.int64, device='cuda')
        _lib.lib.sf_debug_trace(trace.data_ptr())
        run()
        torch.cuda.synchronize()
        _lib.lib.sf_debug_trace(None)
    raw = trace.cpu().numpy().reshape(256, 32).astype(np.float64)
    tr = raw[:, :16] / 100.0 / S
    blk = np.arange(256)
    text = (blk & 7) < 4
    names = ['first product (t_text | q)', 'h stage', 'text attn | visual', 'h~', 'r', 'scores + glue', 'u stages',
             'feature stages', 'tile stores', 'wait partial tiles', 'cell', 'loop back',
             'scores: loads issued', 'scores: parked on r', 'scores: rows + r landed, dots', 'scores: glue']
    print('  mean us per step and phase')
    for k, n in enumerate(names):
        print('    %-28s text %6.2f   visual %6.2f' % (n, tr[text, k].mean(), tr[~text, k].mean()))
    ab = raw[:, 16:32] / 100.0
    t0 = ab[ab > 0].min()
    print('  step %d: phase END times (us after the earliest stamp), min / mean / max over workgroups' % (S // 2))
    order = [11, 0, 1, 2, 3, 4, 12, 13, 14, 15, 5, 6, 7, 8, 9, 10]
    for k in order:
        for lab, sel in (('text', text), ('visual', ~text)):
            v = ab[sel, k]
            v = v[v > 0] - t0
            if len(v):
                print('    %-28s %-5s %7.2f %7.2f %7.2f' % (names[k], lab, v.min(), v.mean(), v.max()))
