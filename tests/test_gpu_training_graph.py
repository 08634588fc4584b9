"""GPU: a whole training iteration as ONE hipGraph (runtime.TrainingGraph).

What differs between two iterations -- dropout masks, sampled actions / words, Adam's bias corrections -- is read from
device words the host writes in front of each replay (sf_dropout.site_dev, sf_sample.stream_dev, sf_adam_step_dev), numbered
exactly like the eager loop.  So N replays must leave the weights N eager iterations leave: same losses, same actions,
same weights to fp32 roundoff of the (non-associative only across streams) accumulation order -- here: bit for bit on one
stream, 1e-6 relative with the two-stream backward.  And the advisor's round-4 finding: a captured INFERENCE pass with
`sample` feedback draws new words / actions on every replay."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speaker_follower_amd import synth                                # noqa: E402


def _follower(seed=21):
    from speaker_follower_amd import model
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return enc.cuda().train(), dec.cuda().train()


def _weights(mods):
    return torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()]).clone()


@pytest.mark.parametrize('feedback,two_stream', [('sample', False), ('teacher', True), ('argmax', True)])
def test_follower_training_graph_equals_the_eager_loop(feedback, two_stream):
    from speaker_follower_amd import features, follower as fol, optim
    B, S, NVP, N = 48, 6, 96, 5
    fb = synth.follower_batch(seed=3, batch=B, steps=S, n_viewpoints=NVP, min_len=8, max_len=40)
    store = features.FeatureStore(synth.feature_table(3, NVP))
    batch = fol.DeviceFollowerBatch.from_synth(fb)
    out = {}
    for mode in ('eager', 'graph'):
        enc, dec = _follower()
        oe = optim.FusedAdam([p for p in enc.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-4)
        od = optim.FusedAdam([p for p in dec.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-4)
        eng = fol.FollowerEngine(enc, dec, store)
        eng.dropout_seed = 777
        eng.two_stream_backward = two_stream
        losses, acts, sites = [], [], []
        if mode == 'eager':
            for _ in range(N):
                oe.zero_grad()
                od.zero_grad()
                st = eng.rollout(batch, S, feedback, train=True)
                st.loss.backward()
                oe.step()
                od.step()
                losses.append(float(st.loss.detach()))
                acts.append(st.actions.cpu().numpy().copy())
                sites.append(st.site0)
        else:
            tg = eng.capture_training(batch, S, feedback, optimizers=(oe, od))      # (runs iteration 1 eagerly)
            losses.append(float(tg.first.loss_buf))
            acts.append(tg.first.actions.cpu().numpy().copy())
            sites.append(tg.first.site0)
            for _ in range(N - 1):
                st = tg.replay()
                torch.cuda.synchronize()
                losses.append(float(st.loss_buf))
                acts.append(st.actions.cpu().numpy().copy())
                sites.append(st.site0)
            assert tg.replays == N - 1 and eng.iteration == N
        torch.cuda.synchronize()
        out[mode] = (losses, acts, sites, _weights((enc, dec)), oe.host_steps() + od.host_steps())
    le, lg = out['eager'][0], out['graph'][0]
    print('[training graph, %s] losses eager %s | graph %s' % (feedback, ['%.5f' % x for x in le], ['%.5f' % x for x in lg]))
    assert out['eager'][2] == out['graph'][2]                      # the same sites ...
    assert out['eager'][4] == out['graph'][4] == [N, N]            # ... and Adam steps
    assert len(set(le)) == N                                       # (iterations differ: fresh masks, moving weights)
    for a, b in zip(out['eager'][1], out['graph'][1]):
        assert np.array_equal(a, b)                                # same sampled / chosen actions
    np.testing.assert_allclose(lg, le, rtol=2e-6)
    we, wg = out['eager'][3], out['graph'][3]
    rel = float((we - wg).abs().max()) / float(we.abs().max())
    print('[training graph, %s] max weight difference after %d iterations: %.2e of the largest weight' % (feedback, N, rel))
    assert rel <= (0.0 if not two_stream else 2e-6)


def test_speaker_training_graph_equals_the_eager_loop():
    from speaker_follower_amd import model, features, speaker, optim
    d = synth.FULL
    N, B, S = 4, 40, 24
    sb = synth.speaker_batch(seed=5, batch=B, n_viewpoints=96, min_len=5, max_len=S - 1)
    store = features.FeatureStore(synth.feature_table(8, 96))
    batch = speaker.DeviceSpeakerBatch.from_synth(sb)
    out = {}
    for mode in ('eager', 'graph'):
        senc_w, sdec_w = synth.speaker_weights_peaky(404)
        enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
        dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
        enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
        dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
        enc.cuda().train()
        dec.cuda().train()
        oe = optim.FusedAdam([p for p in enc.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-4)
        od = optim.FusedAdam([p for p in dec.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-4)
        eng = speaker.SpeakerEngine(enc, dec, store)
        eng.dropout_seed = 4321
        losses = []
        if mode == 'eager':
            for _ in range(N):
                oe.zero_grad()
                od.zero_grad()
                st = eng.score(batch, S, 'teacher', train=True)
                st.loss.backward()
                oe.step()
                od.step()
                losses.append(float(st.loss.detach()))
        else:
            tg = eng.capture_training(batch, S, optimizers=(oe, od))
            losses.append(float(tg.first.loss_buf))
            for _ in range(N - 1):
                st = tg.replay()
                torch.cuda.synchronize()
                losses.append(float(st.loss_buf))
        torch.cuda.synchronize()
        out[mode] = (losses, _weights((enc, dec)))
    print('[speaker training graph] losses eager %s | graph %s' % (out['eager'][0], out['graph'][0]))
    assert len(set(out['eager'][0])) == N
    np.testing.assert_allclose(out['graph'][0], out['eager'][0], rtol=2e-6)
    we, wg = out['eager'][1], out['graph'][1]
    assert float((we - wg).abs().max()) <= 2e-6 * float(we.abs().max())


def test_captured_sampled_passes_draw_anew_on_every_replay():
    """Advisor (round 4): SpeakerEngine.capture / FollowerEngine.capture baked the sampling stream into the graph -- every
    replay with `sample` feedback drew the same words.  Now the stream is a device word: replays differ from each other
    and equal the eager passes with the same host-side site numbers."""
    from speaker_follower_amd import model, features, speaker, follower as fol
    d = synth.FULL
    senc_w, sdec_w = synth.speaker_weights(11)
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    sb = synth.speaker_batch(seed=5, batch=32, n_viewpoints=64, min_len=5, max_len=20)
    store = features.FeatureStore(synth.feature_table(8, 64))
    batch = speaker.DeviceSpeakerBatch.from_synth(sb)
    eng = speaker.SpeakerEngine(enc, dec, store)
    eng.dropout_seed = 99
    replay, st = eng.capture(batch, 12, 'sample')
    words, sites = [], []
    for _ in range(3):
        replay()
        torch.cuda.synchronize()
        words.append(st.words.cpu().numpy().copy())
        sites.append(st.site0)
    assert not np.array_equal(words[0], words[1]) and not np.array_equal(words[1], words[2])
    ref = speaker.SpeakerEngine(enc, dec, store)
    ref.dropout_seed = 99
    for w, site in zip(words, sites):
        ref.site_next = site
        with torch.no_grad():
            e = ref.score(batch, 12, 'sample', train=False)
        assert np.array_equal(e.words.cpu().numpy(), w)
    # the follower's captured rollout with `sample` feedback
    enc_f, dec_f = _follower()
    enc_f.eval()
    dec_f.eval()
    fb = synth.follower_batch(seed=3, batch=32, steps=6, n_viewpoints=64, min_len=8, max_len=30)
    fbatch = fol.DeviceFollowerBatch.from_synth(fb)
    feng = fol.FollowerEngine(enc_f, dec_f, store)
    feng.dropout_seed = 5
    freplay, fst = feng.capture(fbatch, 6, 'sample')
    acts, fsites = [], []
    for _ in range(3):
        freplay()
        torch.cuda.synchronize()
        acts.append(fst.actions.cpu().numpy().copy())
        fsites.append(fst.site0)
    assert not np.array_equal(acts[0], acts[1]) and not np.array_equal(acts[1], acts[2])
    fref = fol.FollowerEngine(enc_f, dec_f, store)
    fref.dropout_seed = 5
    for a, site in zip(acts, fsites):
        fref.site_next = site
        with torch.no_grad():
            e = fref.rollout(fbatch, 6, 'sample', train=False)
        assert np.array_equal(e.actions.cpu().numpy(), a)


def test_adam_step_with_a_device_side_counter():
    """sf_adam_step_dev == sf_adam_step for the same step numbers (bias corrections formed on the device in double)."""
    from speaker_follower_amd import optim
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(1000, device='cuda')), torch.nn.Parameter(torch.randn(37, 5, device='cuda'))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = optim.FusedAdam(a, lr=1e-3, weight_decay=5e-4)
    ob = optim.FusedAdam(b, lr=1e-3, weight_decay=5e-4)
    word = torch.zeros(1, dtype=torch.int32, device='cuda')
    ob.bind_device_steps([word])
    for it in range(4):
        g = [torch.randn_like(p) for p in a]
        for p, q, gg in zip(a, b, g):
            p.grad = gg.clone()
            q.grad = gg.clone()
        oa.step()
        ob.step()
    torch.cuda.synchronize()
    assert int(word) == 4 and ob.host_steps() == [4]
    for p, q in zip(a, b):
        assert torch.equal(p.detach(), q.detach())



def test_guarded_adam_step_skips_while_the_fault_word_is_set():
    """optim.FusedAdam.guard_faults: a device-side step is a no-op (parameters, moments, counter untouched) while the
    fault word of the current stream's workspace is non-zero, and an ordinary step once it is cleared."""
    from speaker_follower_amd import optim, runtime
    torch.manual_seed(1)
    p = [torch.nn.Parameter(torch.randn(515, device='cuda'))]
    q = [torch.nn.Parameter(p[0].detach().clone())]
    guarded, plain = optim.FusedAdam(p, lr=1e-3, weight_decay=5e-4), optim.FusedAdam(q, lr=1e-3, weight_decay=5e-4)
    word, word_q = (torch.zeros(1, dtype=torch.int32, device='cuda') for _ in range(2))
    guarded.bind_device_steps([word])
    plain.bind_device_steps([word_q])
    guarded.guard_faults = True
    fw = runtime.fault_word(torch.device('cuda', 0))
    assert int(fw) == 0
    try:
        for it in range(4):
            g = torch.randn_like(p[0])
            p[0].grad, q[0].grad = g.clone(), g.clone()
            before = p[0].detach().clone()
            if it == 2:
                fw.fill_(runtime.FAULT_ENC_BWD)                     # a starved backward launch raised its bit
                guarded.step()
                torch.cuda.synchronize()
                assert torch.equal(p[0].detach(), before) and int(word) == 2
                fw.zero_()
                guarded.set_host_steps([2])                         # (what the host does when it sees the fault)
            guarded.step()
            plain.step()
        torch.cuda.synchronize()
    finally:
        fw.zero_()
    assert int(word) == 4 and torch.equal(p[0].detach(), q[0].detach())


def test_small_tensors_allocated_after_the_capture_survive_replays():
    """Advisor, round 5: the optimizers' `coef` scratch is a frozen kernel argument of the captured Adam launches.  It
    must stay allocated as long as the graph: blocks handed out by the caching allocator AFTER the capture may never
    lie under the words each replay writes, and the replays must keep stepping with their own coefficients."""
    from speaker_follower_amd import features, follower as fol, optim
    B, S, NVP = 16, 4, 64
    fb = synth.follower_batch(seed=4, batch=B, steps=S, n_viewpoints=NVP, min_len=6, max_len=20)
    store = features.FeatureStore(synth.feature_table(4, NVP))
    batch = fol.DeviceFollowerBatch.from_synth(fb)
    enc, dec = _follower()
    oe = optim.FusedAdam([p for p in enc.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-4)
    od = optim.FusedAdam([p for p in dec.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-4)
    eng = fol.FollowerEngine(enc, dec, store)
    tg = eng.capture_training(batch, S, 'teacher', optimizers=(oe, od))
    scratch = [t for o in (oe, od) for t in o.device_scratch()]
    assert len(scratch) == 2 and all(t is not None for t in scratch)          # still owned after the capture ended
    held = {t.data_ptr() for t in scratch}
    torch.cuda.synchronize()
    # many small blocks of the size class the scratch came from: none may alias it
    small = [torch.full((4,), float(i), device='cuda') for i in range(4096)]
    assert not held & {t.data_ptr() for t in small}
    w0 = _weights((enc, dec))
    for _ in range(3):
        tg.replay()
    torch.cuda.synchronize()
    got = torch.stack(small)
    assert torch.equal(got, torch.arange(4096, device='cuda', dtype=torch.float32)[:, None].expand(4096, 4))
    w3 = _weights((enc, dec))
    assert torch.isfinite(w3).all() and float((w3 - w0).abs().max()) > 0          # the replays stepped
    # the step sizes the replays used are Adam's for steps 2..4 (bias corrections of THIS optimizer): |dw| <= lr * ~1
    assert float((w3 - w0).abs().max()) < 3 * 1e-3 * 1.5
