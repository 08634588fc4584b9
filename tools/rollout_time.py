"""Headline rollout (B = 100, 20 steps, full table): hipGraph replay time + the per-kernel table of one eager rollout."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower
from speaker_follower_amd._lib import kernel_profile
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
enc.eval(); dec.eval()
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=10567)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
eng = follower.FollowerEngine(enc, dec, store)
eng.fold_inference = '--fold' in os.environ.get('ROLLOUT_FLAGS', '')          # ROLLOUT_FLAGS=--fold: the folded inference schedule
replay, st = eng.capture(batch, 20, 'argmax')
best = 1e9
for rnd in range(4):
    for _ in range(10):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    best = min(best, dt)
    print('rollout %.4f ms (%.0f agent-steps/s)' % (1e3 * dt, 2000 / dt))
with torch.no_grad():
    eng.rollout(batch, 20, 'argmax', train=False)
    with kernel_profile() as prof:
        for _ in range(3):
            eng.rollout(batch, 20, 'argmax', train=False)
torch.cuda.synchronize()
tot = sum(v['total_us'] for v in prof.rows.values()) / 3
print('kernel time per rollout %.1f us' % tot)
for k, v in sorted(prof.rows.items(), key=lambda kv: -kv[1]['total_us'])[:14]:
    print('  %-60s %5.1f calls %7.2f us avg %7.1f us/rollout' % (k[:60], v['calls'] / 3, v['avg_us'], v['total_us'] / 3))
print('checksum actions', int(st.actions.sum()))
