"""Teacher-forced speaker passes: batched form (round 5) vs the persistent word loop, and the per-kernel table of the
training iteration."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, speaker, bench_extras, optim
dev = torch.device('cuda', 0)
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
enc, dec = bench_extras._speaker_models(dev)
sb = synth.speaker_batch(seed=0, batch=100, n_viewpoints=10567, min_path=4, max_path=7, min_len=10, max_len=79)
b = speaker.DeviceSpeakerBatch.from_synth(sb, device=dev)
for batched in (False, True):
    eng = speaker.SpeakerEngine(enc, dec, store)
    eng.teacher_batched = batched
    replay, st = eng.capture(b, 80, 'teacher')
    dt = bench_extras._timed(replay, 5, 20)
    print('teacher scoring 100 x 80, %s: %.3f ms per batch (graph replay)' % ('batched head + persistent recurrence' if batched else 'persistent word loop', 1e3 * dt))
    def eager():
        with torch.no_grad():
            eng.score(b, 80, 'teacher', train=False)
    rows, us = bench_extras.kernel_table(eager)
    for r in rows[:14]:
        print('   %-70s %6.1f calls %8.1f us avg %8.1f us/run %5.1f%%' % (r['kernel'][:70], r['calls_per_run'], r['avg_us'], r['us_per_run'], 100 * r['share']))
enc.train(); dec.train()
pe = [p for p in enc.parameters() if p.requires_grad]; pd = [p for p in dec.parameters() if p.requires_grad]
oe, od = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4), optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
eng = speaker.SpeakerEngine(enc, dec, store)
def it():
    oe.zero_grad(); od.zero_grad()
    st = eng.score(b, 80, 'teacher', train=True)
    st.loss.backward()
    oe.step(); od.step()
rows, us = bench_extras.kernel_table(it, reps=3, top=40)
print('training iteration: %.3f ms of kernels' % (1e-3 * us))
for r in rows:
    print('   %-70s %6.1f calls %8.1f us avg %8.1f us/run %5.1f%%' % (r['kernel'][:70], r['calls_per_run'], r['avg_us'], r['us_per_run'], 100 * r['share']))
