"""GPU: the persistent decode loop (csrc/sf_mega.hip), milestone by milestone, against the tapes of the
per-stage engine (which is itself pinned to the reference's goldens)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speaker_follower_amd import synth                                # noqa: E402


def reference_rollout(B, S, seed=47, peaky=True):
    from speaker_follower_amd import model, features, follower
    d = synth.FULL
    enc_w, dec_w = (synth.follower_weights_peaky if peaky else synth.follower_weights)(303)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    fb = synth.follower_batch(seed=seed, batch=B, steps=S, n_viewpoints=256)
    store = features.FeatureStore(synth.feature_table(8, 256))
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    with torch.no_grad():
        st = follower.FollowerEngine(enc, dec, store).rollout(batch, S, 'argmax', train=False)
    torch.cuda.synchronize()
    return enc, dec, store, batch, st


@pytest.mark.parametrize('B,S', [(100, 20), (128, 5), (16, 6), (37, 9)])
def test_milestone1_lstm_loop_matches_the_step_kernels(B, S):
    """Gate product + cell + h feedback of all S decode steps in one launch, (u | feature) from the
    reference tape: h1 / c1 of every step as the gemm_nt_tiled + lstm_pw pair computes them."""
    from speaker_follower_amd import _lib
    from speaker_follower_amd.model import decoder_params
    from speaker_follower_amd.runtime import ptr, ws_args
    enc, dec, store, batch, st = reference_rollout(B, S)
    p = decoder_params(dec)
    lw = _lib.LstmW(p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(), p[3].data_ptr(), None, None)
    H = 512
    for rep in range(3):
        h1 = torch.full((S, B, H), float('nan'), device='cuda')
        c1 = torch.full((S, B, H), float('nan'), device='cuda')
        _lib.call('sf_debug_mega_lstm_loop', C.byref(lw), ptr(st.h_init), ptr(st.c_init), ptr(st.tape['xin']), B, S,
                  ptr(h1), ptr(c1), None, *ws_args(h1.device))
        torch.cuda.synchronize()
        assert not torch.isnan(h1).any()
        torch.testing.assert_close(h1, st.tape['h1'], rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(c1, st.tape['c1'], rtol=2e-5, atol=2e-6)


def test_milestone1_timing():
    from speaker_follower_amd import _lib
    from speaker_follower_amd.model import decoder_params
    from speaker_follower_amd.runtime import ptr, ws_args
    B, S = 100, 20
    enc, dec, store, batch, st = reference_rollout(B, S)
    p = decoder_params(dec)
    lw = _lib.LstmW(p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(), p[3].data_ptr(), None, None)
    h1 = torch.empty(S, B, 512, device='cuda')
    c1 = torch.empty(S, B, 512, device='cuda')
    run = lambda: _lib.call('sf_debug_mega_lstm_loop', C.byref(lw), ptr(st.h_init), ptr(st.c_init), ptr(st.tape['xin']),  # noqa: E731
                            B, S, ptr(h1), ptr(c1), None, *ws_args(h1.device))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    with _lib.kernel_profile() as prof:
        for _ in range(5):
            run()
    print({k: round(v['avg_us'], 1) for k, v in prof.rows.items()})
    mk = [v for k, v in prof.rows.items() if 'mega_kernel' in k][0]
    print('gate product + cell, %d steps in one launch: %.1f us = %.2f us per step (per-stage kernels: 25.0 + 4.3)'
          % (S, mk['avg_us'], mk['avg_us'] / S))
