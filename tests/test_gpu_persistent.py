"""GPU: the persistent encoder launch (csrc/sf_persist.hip; EncoderLSTM.forward, model.py:81-104)
against the one-launch-per-step path -- for ragged lengths, batch sizes that leave row groups partly or
wholly empty, train-mode dropout on ctx, repeated launches and launches racing on two streams.  Rounds 2-3:
same fp32-MFMA summation order, BIT-identical.  Round 4: the persistent launch forms its products on the bf16
matrix cores with error-free operand splitting (csrc/sf_split.h: fp32 accuracy class, fewer roundings), so the
two paths agree to fp32 roundoff -- every tape within 3e-6 over up to 80 recurrent steps -- and both are checked
against a float64 evaluation of the recurrence: the persistent launch must not be further from it than the
per-step kernels are.  Repeated launches of the persistent kernel still reproduce their own bits."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speaker_follower_amd import synth                                # noqa: E402


def encoder(seed=101):
    from speaker_follower_amd import model
    d = synth.FULL
    enc_w, _ = synth.follower_weights(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    return enc.cuda()


def batch(seed, B, min_len, max_len):
    from speaker_follower_amd.follower import batch_instructions_from_encoded
    instr = synth.instructions(seed, B, min_len, max_len, synth.FULL, sort=True)
    return batch_instructions_from_encoded(instr, 80, reverse=True)


def run(enc, seq, lens, persistent, train=False):
    from speaker_follower_amd import _lib
    from speaker_follower_amd.model import _encoder_structs
    from speaker_follower_amd.runtime import ptr, ws_args, dropout_arg
    import ctypes as C
    enc.persistent = persistent
    B, Lpad = seq.shape
    T, E, H = max(lens), enc.embedding_size, enc.hidden_size
    new = lambda *s: torch.full(s, float('nan'), device='cuda', dtype=torch.float32)   # noqa: E731
    out = dict(ctx=new(B, T, H), h=new(B, H), c=new(B, H), emb=new(T, B, E), xg=new(T, B, 4 * H),
               gates=new(T, B, 4 * H), hs=new(T + 1, B, H), cs=new(T + 1, B, H))
    tp = _lib.EncoderTape(*(out[k].data_ptr() for k in ('emb', 'xg', 'gates', 'hs', 'cs')))
    w = _encoder_structs(enc)
    lens_dev = torch.tensor(lens, dtype=torch.int32, device='cuda')
    _lib.call('sf_encoder_lstm_fwd', C.byref(w), B, Lpad, T, E, H, ptr(seq), ptr(lens_dev), ptr(out['ctx']),
              ptr(out['h']), ptr(out['c']), C.byref(tp), dropout_arg(0.5 if train else 0.0, 0xBEEF, 3), 7,
              *ws_args(seq.device))
    out.pop('xg')            # the per-step path with a table never writes it either
    return out


def _encoder_f64(enc, seq, lens):
    """The recurrence in float64 on the host (model.py:81-104, eval mode): hs [T+1,B,H], cs."""
    w = {k: v.detach().double().cpu() for k, v in enc.state_dict().items()}
    B, T, H = seq.shape[0], max(lens), enc.hidden_size
    x = w['embedding.weight'][seq.cpu()[:, :T]]                       # [B,T,E]
    h = torch.zeros(B, H, dtype=torch.float64)
    c = torch.zeros(B, H, dtype=torch.float64)
    hs, cs = [h], [c]
    ln = torch.tensor(lens)
    for t in range(T):
        g = x[:, t] @ w['lstm.weight_ih_l0'].T + w['lstm.bias_ih_l0'] + h @ w['lstm.weight_hh_l0'].T + w['lstm.bias_hh_l0']
        i, f, gg, o = g.chunk(4, 1)
        c1 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h1 = torch.sigmoid(o) * torch.tanh(c1)
        live = (t < ln)[:, None]
        h, c = torch.where(live, h1, h), torch.where(live, c1, c)
        hs.append(h)
        cs.append(c)
    return torch.stack(hs), torch.stack(cs)


@pytest.mark.parametrize('B,min_len,max_len,train', [(100, 10, 79, False), (100, 10, 79, True), (8, 3, 19, False),
                                                     (13, 1, 5, False), (128, 2, 79, False), (97, 79, 79, True),
                                                     (1, 4, 4, False)])
def test_persistent_encoder_matches_per_step_and_float64(B, min_len, max_len, train):
    enc = encoder()
    seq, mask, lens = batch(B + 1, B, min_len, max_len)
    ref = run(enc, seq, lens, persistent=False, train=train)
    got = run(enc, seq, lens, persistent=True, train=train)
    torch.cuda.synchronize()
    for k in ref:
        assert not torch.isnan(ref[k]).any() and not torch.isnan(got[k]).any(), k
        torch.testing.assert_close(got[k], ref[k], rtol=0, atol=3e-6)
    hs64, cs64 = _encoder_f64(enc, seq, lens)
    e_per = float((ref['hs'].double().cpu() - hs64).abs().max())
    e_pst = float((got['hs'].double().cpu() - hs64).abs().max())
    print('[encoder B=%d T=%d] max|h - float64|: per-step fp32 MFMA %.2e, persistent bf16 x 6 %.2e' % (B, max(lens), e_per, e_pst))
    assert e_pst <= 2e-6 and e_pst <= 1.5 * e_per + 2e-7


def test_persistent_encoder_kernel_is_the_one_that_runs():
    """The default path at the headline shape is the persistent launch (not a silent fallback)."""
    from speaker_follower_amd import _lib
    enc = encoder()
    seq, mask, lens = batch(5, 100, 10, 79)
    run(enc, seq, lens, persistent=True)
    torch.cuda.synchronize()
    with _lib.kernel_profile() as prof:
        run(enc, seq, lens, persistent=True)
    names = ' '.join(prof.rows)
    assert 'enc_persist_kernel' in names and 'lstm_step' not in names, names
    with _lib.kernel_profile() as prof:
        run(enc, seq, lens, persistent=False)
    assert 'enc_persist_kernel' not in ' '.join(prof.rows)


def test_persistent_encoder_soak_and_two_streams():
    """200 back-to-back launches and two streams racing for the device-wide lock give the same bits
    every time (an exchange that trusted stale data or a torn reset would not)."""
    enc = encoder(3)
    seq, mask, lens = batch(9, 100, 10, 79)
    ref = run(enc, seq, lens, persistent=True)
    torch.cuda.synchronize()
    per_step = run(enc, seq, lens, persistent=False)
    torch.testing.assert_close(ref['ctx'], per_step['ctx'], rtol=0, atol=3e-6)
    for _ in range(100):
        got = run(enc, seq, lens, persistent=True)
    torch.cuda.synchronize()
    assert torch.equal(got['ctx'], ref['ctx']) and torch.equal(got['c'], ref['c'])
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for i in range(50):
        for s in (s1, s2):
            with torch.cuda.stream(s):
                outs.append(run(enc, seq, lens, persistent=True))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o['ctx'], ref['ctx']) and torch.equal(o['h'], ref['h'])


def test_larger_batches_fall_back_to_the_per_step_path():
    enc = encoder()
    seq, mask, lens = batch(2, 160, 5, 40)
    a = run(enc, seq, lens, persistent=True)
    b = run(enc, seq, lens, persistent=False)
    torch.cuda.synchronize()
    assert torch.equal(a['ctx'], b['ctx'])


def run_bwd(enc, seq, lens, fwd, persistent, train, dctx, d_init, d_ct):
    """sf_encoder_lstm_bwd on a copy of the forward tape `fwd`; returns (dgates tape, weight grads)."""
    from speaker_follower_amd import _lib
    from speaker_follower_amd.model import _encoder_structs
    from speaker_follower_amd.runtime import ptr, ws_args, dropout_arg
    import ctypes as C
    enc.persistent = persistent
    B, Lpad = seq.shape
    T, E, H = max(lens), enc.embedding_size, enc.hidden_size
    tape = {k: fwd[k].clone() for k in ('emb', 'gates', 'hs', 'cs')}
    tape['xg'] = torch.zeros(T, B, 4 * H, device='cuda')
    tp = _lib.EncoderTape(*(tape[k].data_ptr() for k in ('emb', 'xg', 'gates', 'hs', 'cs')))
    for p in enc.parameters():
        p.grad = None
    w, g = _encoder_structs(enc), _encoder_structs(enc, grad=True)
    lens_dev = torch.tensor(lens, dtype=torch.int32, device='cuda')
    _lib.call('sf_encoder_lstm_bwd', C.byref(w), C.byref(g), B, T, E, H, ptr(lens_dev), ptr(fwd['h']), ptr(dctx),
              ptr(d_init), ptr(d_ct), C.byref(tp), dropout_arg(0.5 if train else 0.0, 0xBEEF, 3), 7,
              *ws_args(seq.device))
    torch.cuda.synchronize()
    return tape['xg'], {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('B,min_len,max_len,train', [(100, 10, 79, True), (100, 79, 79, False), (13, 1, 5, False),
                                                     (128, 2, 60, True), (40, 3, 33, False)])
def test_persistent_encoder_backward_matches_per_step(B, min_len, max_len, train):
    """The backward recurrence as one launch (enc_bwd_persist_kernel) against lstm_bwd_step_fused per
    step: the same dgates tape and weight gradients (different summation order: 1e-5 of the scale)."""
    enc = encoder(11)
    seq, mask, lens = batch(B + 7, B, min_len, max_len)
    fwd = run(enc, seq, lens, persistent=True, train=train)
    fwd['xg'] = None
    T, H = max(lens), enc.hidden_size
    g = torch.Generator(device='cuda').manual_seed(B)
    dctx = torch.randn(B, T, H, device='cuda', generator=g)
    d_init = torch.randn(B, H, device='cuda', generator=g)
    d_ct = torch.randn(B, H, device='cuda', generator=g)
    ref_dg, ref_gr = run_bwd(enc, seq, lens, fwd, False, train, dctx, d_init, d_ct)
    for _ in range(3):                                   # repeated launches reuse the self-resetting buffers
        got_dg, got_gr = run_bwd(enc, seq, lens, fwd, True, train, dctx, d_init, d_ct)
    assert not torch.isnan(got_dg).any()
    scale = float(ref_dg.abs().max())
    assert float((got_dg - ref_dg).abs().max()) <= 1e-5 * scale
    assert set(got_gr) == set(ref_gr) and len(ref_gr) >= 5
    for k in ref_gr:
        s = float(ref_gr[k].abs().max())
        assert float((got_gr[k] - ref_gr[k]).abs().max()) <= 2e-5 * s, k


def test_persistent_backward_is_the_kernel_that_runs():
    from speaker_follower_amd import _lib
    enc = encoder()
    seq, mask, lens = batch(5, 100, 10, 79)
    fwd = run(enc, seq, lens, persistent=True)
    T, H = max(lens), enc.hidden_size
    dctx = torch.randn(100, T, H, device='cuda')
    z = torch.zeros(100, H, device='cuda')
    run_bwd(enc, seq, lens, fwd, True, False, dctx, z, z)
    with _lib.kernel_profile() as prof:
        run_bwd(enc, seq, lens, fwd, True, False, dctx, z, z)
    names = ' '.join(prof.rows)
    assert 'enc_bwd_persist_kernel' in names and 'lstm_bwd_step' not in names, names
    print({k: round(v['avg_us'], 1) for k, v in prof.rows.items() if 'persist' in k}, 'T =', T)


# ---------------------------------------------------------------------------------- speaker word loop
def speaker_setup(B, seed=404, peaky=True):
    from speaker_follower_amd import model, features, speaker
    d = synth.FULL
    senc_w, sdec_w = (synth.speaker_weights_peaky if peaky else synth.speaker_weights)(seed)
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
    enc.cuda().eval()
    dec.cuda().eval()
    sb = synth.speaker_batch(seed=B + 1, batch=B, n_viewpoints=128, min_len=5, max_len=60)
    store = features.FeatureStore(synth.feature_table(8, 128))
    return enc, dec, store, speaker.DeviceSpeakerBatch.from_synth(sb)


@pytest.mark.parametrize('B,feedback,peaky', [(100, 'argmax', True), (100, 'teacher', True), (37, 'argmax', False),
                                              (128, 'teacher', False), (5, 'argmax', True)])
def test_persistent_speaker_decode_matches_per_step(B, feedback, peaky):
    """sf_speaker_decode (one launch for S word steps, folded attention) against S x
    (sf_speaker_decoder_fwd + sf_speaker_glue_fwd): identical words, logits within 1e-4 of their
    scale, scores / NLL terms / attention / final state to 1e-4."""
    from speaker_follower_amd import speaker
    enc, dec, store, batch = speaker_setup(B, peaky=peaky)
    S = 40
    out = []
    for persistent in (False, True):
        eng = speaker.SpeakerEngine(enc, dec, store)
        eng.persistent = persistent
        eng.teacher_batched = False          # (teacher mode of sf_speaker_decode itself; the batched form: test_gpu_speaker_teacher.py)
        with torch.no_grad():
            st = eng.score(batch, S, feedback, train=False)
        assert st.persistent == persistent
        out.append(st)
    a, b = out
    torch.cuda.synchronize()
    assert torch.equal(a.words, b.words)
    la, lb = a.logits.cpu().numpy(), b.logits.cpu().numpy()
    assert float(np.abs(la - lb).max()) <= 1e-4 * float(np.abs(la).max())
    np.testing.assert_allclose(b.step_scores.cpu().numpy(), a.step_scores.cpu().numpy(), rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(b.nll_term.cpu().numpy(), a.nll_term.cpu().numpy(), rtol=1e-4, atol=2e-4)
    assert torch.equal(a.live, b.live) and torch.equal(a.ended, b.ended)
    np.testing.assert_allclose(b.tape['alpha'].cpu().numpy(), a.tape['alpha'].cpu().numpy(), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(b.h.cpu().numpy(), a.h.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b.c.cpu().numpy(), a.c.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(b.loss), float(a.loss), rtol=1e-5)


def test_persistent_speaker_decode_timing_and_graph():
    from speaker_follower_amd import speaker, _lib
    enc, dec, store, batch = speaker_setup(100)
    res = {}
    for persistent in (False, True):
        eng = speaker.SpeakerEngine(enc, dec, store)
        eng.persistent = persistent
        replay, gst = eng.capture(batch, 80, 'argmax')
        for _ in range(3):
            replay()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(10):
            replay()
        torch.cuda.synchronize()
        res[persistent] = ((time.perf_counter() - t0) / 10, gst.words.clone())
    assert torch.equal(res[True][1], res[False][1])
    print('speaker 100 x 80 greedy words: per-step %.2f ms, persistent %.2f ms' % (1e3 * res[False][0], 1e3 * res[True][0]))
    with torch.no_grad():
        eng = speaker.SpeakerEngine(enc, dec, store)
        eng.score(batch, 80, 'argmax', train=False)
        with _lib.kernel_profile() as prof:
            eng.score(batch, 80, 'argmax', train=False)
    print({k: round(v['avg_us'], 1) for k, v in prof.rows.items() if 'persist' in k})
    assert res[True][0] < res[False][0]


def test_placement_independent_exchange_gives_the_same_results():
    """The persistent launches take a fast path (plain stores, exchange inside one XCD's L2) when the
    hardware XCC ids of a row group agree; otherwise write-through stores make the protocol valid for
    any placement.  Force the fallback: forward, backward and the speaker word loop bit-identical to the
    fast path."""
    from speaker_follower_amd import _lib, speaker
    enc = encoder(5)
    seq, mask, lens = batch(21, 100, 10, 79)
    T, H = max(lens), enc.hidden_size
    ref = run(enc, seq, lens, persistent=True)                   # the fast (one-XCD) exchange
    dctx = torch.randn(100, T, H, device='cuda')
    z = torch.zeros(100, H, device='cuda')
    fast_dg, _ = run_bwd(enc, seq, lens, ref, True, False, dctx, z, z)
    senc, sdec, store, sbatch = speaker_setup(100)
    with torch.no_grad():
        fast_spk = speaker.SpeakerEngine(senc, sdec, store).score(sbatch, 30, 'argmax', train=False)
    _lib.lib.sf_debug_force_write_through(1)
    try:
        trace = torch.zeros(256 * 8, dtype=torch.int64, device='cuda')
        _lib.lib.sf_debug_trace(trace.data_ptr())
        got = run(enc, seq, lens, persistent=True)
        torch.cuda.synchronize()
        _lib.lib.sf_debug_trace(None)
        assert int(trace.view(256, 8)[:, 5].sum()) == 0          # no workgroup took the one-XCD path
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
        slow_dg, _ = run_bwd(enc, seq, lens, ref, True, False, dctx, z, z)
        assert torch.equal(slow_dg, fast_dg)
        with torch.no_grad():
            slow_spk = speaker.SpeakerEngine(senc, sdec, store).score(sbatch, 30, 'argmax', train=False)
        assert torch.equal(slow_spk.words, fast_spk.words) and torch.equal(slow_spk.logits, fast_spk.logits)
    finally:
        _lib.lib.sf_debug_trace(None)
        _lib.lib.sf_debug_force_write_through(0)
