"""Soak of the persistent launches after the round-3 changes (sentinel resets behind barriers, per-wave statistics
reset, gate tiles under the exchanges): many replays of the speaker word loop (B = 100 and 128) and of the follower
rollout (persistent encoder) must reproduce the first result bit for bit, with no NaN poison, also with the
placement-independent (write-through) exchange forced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, speaker, bench_extras, _lib
dev = torch.device('cuda', 0)
store = features.FeatureStore(bench.device_table(2048, 1234, dev), device=dev)
N = int(os.environ.get('REPLAYS', 1500))
bad = 0
for force in (0, 1):
    _lib.lib.sf_debug_force_write_through(force)
    for B in (100, 128, 37):
        senc, sdec = bench_extras._speaker_models(dev)
        sb = synth.speaker_batch(seed=B, batch=B, n_viewpoints=2048, min_path=4, max_path=7, min_len=10, max_len=79)
        b = speaker.DeviceSpeakerBatch.from_synth(sb, device=dev)
        eng = speaker.SpeakerEngine(senc, sdec, store)
        replay, st = eng.capture(b, 80, 'argmax')
        replay(); torch.cuda.synchronize()
        w0, s0 = st.words.clone(), st.step_scores.clone()
        assert st.persistent and not torch.isnan(s0).any()
        t0 = time.time()
        for i in range(N):
            replay()
            if i % 100 == 99:
                torch.cuda.synchronize()
                if not (torch.equal(st.words, w0) and torch.equal(st.step_scores, s0)):
                    bad += 1
        torch.cuda.synchronize()
        print('speaker B=%3d force_sc1=%d: %d replays in %.2f s, %s' % (B, force, N, time.time() - t0, 'bit-identical' if bad == 0 else 'MISMATCH'))
    enc, dec, _, _ = bench.build_models(101, dev); enc.eval(); dec.eval()
    fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=2048)
    batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
    replay, st = follower.FollowerEngine(enc, dec, store).capture(batch, 20, 'argmax')
    replay(); torch.cuda.synchronize()
    a0, l0 = st.actions.clone(), st.loss_buf.clone()
    for i in range(N // 3):
        replay()
        if i % 100 == 99:
            torch.cuda.synchronize()
            if not (torch.equal(st.actions, a0) and torch.equal(st.loss_buf, l0)):
                bad += 1
    torch.cuda.synchronize()
    print('follower rollout force_sc1=%d: %d replays, %s' % (force, N // 3, 'bit-identical' if bad == 0 else 'MISMATCH'))
_lib.lib.sf_debug_force_write_through(0)
assert bad == 0
print('soak ok')

# ---- the same under co-tenancy: a second stream keeps the chip busy with large products while the persistent
# launches run (workgroups of a row group then progress with skew: the case the per-wave / behind-the-barrier sentinel
# resets are there for)
senc, sdec = bench_extras._speaker_models(dev)
sb = synth.speaker_batch(seed=5, batch=100, n_viewpoints=2048, min_path=4, max_path=7, min_len=10, max_len=79)
b = speaker.DeviceSpeakerBatch.from_synth(sb, device=dev)
eng = speaker.SpeakerEngine(senc, sdec, store)
replay, st = eng.capture(b, 80, 'argmax')
replay(); torch.cuda.synchronize()
w0, s0 = st.words.clone(), st.step_scores.clone()
side = torch.cuda.Stream()
x = torch.randn(4096, 4096, device=dev)
bad2 = 0
for i in range(300):
    with torch.cuda.stream(side):
        for _ in range(3):
            y = x @ x
    replay()
    if i % 50 == 49:
        torch.cuda.synchronize()
        if not (torch.equal(st.words, w0) and torch.equal(st.step_scores, s0)):
            bad2 += 1
torch.cuda.synchronize()
print('speaker under co-tenancy: 300 replays beside 900 4096^3 products, %s' % ('bit-identical' if bad2 == 0 else '%d MISMATCHES' % bad2))
assert bad2 == 0
