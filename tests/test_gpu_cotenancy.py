"""GPU: the persistent launches beside a COLLECTIVE-SHAPED co-tenant.

Data-parallel training launches the first gradient bucket's all-reduce (40 MB) from the backward while the main stream
still runs `sf_encoder_lstm_bwd` -- since round 2 ONE persistent launch of 256 co-resident workgroups that wait on each
other's data (bounded: 0.25 s, then NaN poison + fault word).  RCCL's reduction kernels are a few dozen workgroups of
256-512 threads with a large LDS allocation that stay on the chip for the length of the transfer; a one-rank RCCL
all-reduce launches none, so `tests/test_gpu_rccl.py` cannot see whether the two starve each other.  This test puts a
kernel of that shape (`sf_debug_cotenant`: 48 workgroups x 512 threads x 64 KB of LDS, resident for 0.5 ms) on its own
stream at exactly the point where `dp.BucketedGrads.launch(0)` would start the collective -- and, for the inference
launches (persistent encoder forward, persistent word loop), at the start of every pass -- for 200 iterations:

  * `persistent_launch_faults` stays 0 and no pass falls back to the per-step kernels;
  * no iteration jumps: the worst iteration is far below the 0.25 s wait bound, the median moves by less than the
    co-tenant's own length;
  * results equal the undisturbed run's (the co-tenant only takes CUs, the exchange protocol does not care).

Why it holds: a persistent launch needs one workgroup per CU to be resident; a co-tenant workgroup that
holds a CU only DELAYS that CU's workgroup until the co-tenant leaves or its resources fit beside it (the persistent
kernels take 4 waves, <= 256 VGPRs per lane and < 64 KB of LDS per CU), and the co-tenant never waits for the persistent
launch -- no cycle, so the bounded waits are only reached if a collective itself stalls for > 0.25 s (a straggling peer),
which the fault word + per-step re-issue then absorbs."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speaker_follower_amd import synth                                # noqa: E402

CO = dict(blocks=48, threads=512, lds=65536, ticks=50000)             # 0.5 ms


class CoTenantSync:
    """Stands where dp.BucketedGrads stands in FollowerEngine._backward: launch(0) starts the co-tenant on its own
    stream behind the current stream's work, the way the process group starts bucket 0's all-reduce."""

    def __init__(self, dev, on=True):
        from speaker_follower_amd.runtime import concurrent_stream
        self.stream = concurrent_stream(dev)
        self.sink = torch.zeros(256, device=dev)
        self.on, self.launched, self.n = on, [], 0

    def launch(self, b):
        from speaker_follower_amd import _lib
        self.launched.append(b)
        if b == 0 and self.on:
            self.stream.wait_stream(torch.cuda.current_stream())
            _lib.call('sf_debug_cotenant', CO['blocks'], CO['threads'], CO['lds'], CO['ticks'],
                      self.sink.data_ptr(), self.stream.cuda_stream)
            self.n += 1

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        self.launched = []

    def abort(self):
        self.wait()

    def start_now(self):
        """(inference passes: beside the pass from its first launch on)"""
        self.launch(0)
        self.launched = []


def _follower(seed=11):
    from speaker_follower_amd import model
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return enc.cuda(), dec.cuda()


def _stats(ms):
    ms = np.asarray(ms)
    return float(np.median(ms)), float(ms.max())


@pytest.mark.parametrize('blocks', [48, 256])
def test_training_iteration_beside_a_collective_shaped_kernel(blocks, monkeypatch):
    """enc_persist_kernel (forward) and enc_bwd_persist_kernel beside the co-tenant launched from the backward:
    48 workgroups (a collective's channels) and 256 (one on EVERY CU: the persistent launch can only wait)."""
    monkeypatch.setitem(CO, 'blocks', blocks)
    from speaker_follower_amd import features, follower as fol, runtime, dp
    dev = torch.device('cuda', 0)
    B, S, NVP, N = 100, 6, 128, 200
    fb = synth.follower_batch(seed=8, batch=B, steps=S, n_viewpoints=NVP, min_len=20, max_len=79)
    store = features.FeatureStore(synth.feature_table(3, NVP))
    batch = fol.DeviceFollowerBatch.from_synth(fb)
    res = {}
    for on in (False, True):
        enc, dec = _follower()
        enc.train()
        dec.train()
        flat = dp.FlatGrads(list(enc.parameters()) + list(dec.parameters()))
        eng = fol.FollowerEngine(enc, dec, store)
        eng.dropout_seed = 4242
        sync = eng.grad_sync = CoTenantSync(dev, on)
        runtime.take_fault(dev)
        ms, faults = [], 0
        for it in range(N + 5):
            flat.zero()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = eng.rollout(batch, S, 'teacher', train=True)
            st.loss.backward()
            sync.wait()
            torch.cuda.synchronize()
            if it >= 5:
                ms.append(1e3 * (time.perf_counter() - t0))
            faults |= runtime.take_fault(dev)
        assert faults == 0 and eng.fallbacks == 0
        assert sync.n == (N + 5 if on else 0)
        assert torch.isfinite(flat.flat).all()
        res[on] = (_stats(ms), float(st.loss), flat.flat.clone())
    (m0, w0), (m1, w1) = res[False][0], res[True][0]
    print('training iteration (B=%d, %d steps): median %.3f ms alone, %.3f ms beside a %d x %d-thread x %d KB co-tenant of '
          '0.5 ms; worst %.2f / %.2f ms; faults 0' % (B, S, m0, m1, CO['blocks'], CO['threads'], CO['lds'] // 1024, w0, w1))
    assert w1 < 60.0                                  # nowhere near a 250 ms bounded wait
    # a few co-tenant lengths at most per iteration (measured: +0.4 ms typically, +1.9 ms on one box in round 6 when the
    # 256-workgroup co-tenant queued in front of BOTH persistent launches) -- the point is the absence of starvation,
    # which would show as the bounded waits' hundreds of milliseconds, not a tight overlap figure
    assert m1 <= m0 + 2.5
    assert res[False][1] == res[True][1]
    assert torch.equal(res[False][2], res[True][2])   # same dropout sites, same bits


@pytest.mark.parametrize('what', ['follower_inference', 'speaker_words'])
def test_inference_passes_beside_a_collective_shaped_kernel(what):
    """enc_persist_kernel / spk_persist_kernel with the co-tenant started at the head of every pass."""
    from speaker_follower_amd import features, follower as fol, runtime, speaker
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_persistent import speaker_setup
    dev = torch.device('cuda', 0)
    N = 200
    if what == 'follower_inference':
        B, S, NVP = 100, 6, 128
        fb = synth.follower_batch(seed=8, batch=B, steps=S, n_viewpoints=NVP, min_len=20, max_len=79)
        enc, dec = _follower()
        enc.eval()
        dec.eval()
        eng = fol.FollowerEngine(enc, dec, features.FeatureStore(synth.feature_table(3, NVP)))
        batch = fol.DeviceFollowerBatch.from_synth(fb)
        run = lambda: eng.rollout(batch, S, 'argmax', train=False)             # noqa: E731
        key = lambda st: (st.actions.clone(), st.logits.clone())               # noqa: E731
    else:
        enc, dec, store, batch = speaker_setup(100)
        eng = speaker.SpeakerEngine(enc, dec, store)
        run = lambda: eng.score(batch, 40, 'argmax', train=False)              # noqa: E731
        key = lambda st: (st.words.clone(), st.logits.clone())                 # noqa: E731
    res = {}
    with torch.no_grad():
        for on in (False, True):
            co = CoTenantSync(dev, on)
            runtime.take_fault(dev)
            ms, faults, st = [], 0, None
            for it in range(N + 5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                co.start_now()
                st = run()
                co.wait()
                torch.cuda.synchronize()
                if it >= 5:
                    ms.append(1e3 * (time.perf_counter() - t0))
                faults |= runtime.take_fault(dev)
            assert faults == 0
            if what == 'speaker_words':
                assert st.persistent
            res[on] = (_stats(ms), key(st))
    (m0, w0), (m1, w1) = res[False][0], res[True][0]
    print('%s: median %.3f ms alone, %.3f ms beside the co-tenant; worst %.2f / %.2f ms; faults 0' % (what, m0, m1, w0, w1))
    assert w1 < 60.0 and m1 <= m0 + 2.5
    for a, b in zip(res[False][1], res[True][1]):
        assert torch.equal(a, b)
