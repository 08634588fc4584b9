"""Per-kernel table of ONE flat decoder step over 64 x 40 = 2 560 search states (what frontier.beam_search issues per step)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import bench_extras, features, synth, search
from speaker_follower_amd.follower import batch_instructions_from_encoded
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
enc.eval(); dec.eval()
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
rng = np.random.default_rng(7)
instances, k = 64, 40
instr = synth.instructions(3, instances, 10, 79, synth.FULL, sort=True)
seq, mask, lengths = batch_instructions_from_encoded(instr, 80, reverse=True, device=dev)
with torch.no_grad():
    ctx, h_t, c_t = enc(seq, lengths)
n = instances * k
obs, udesc = bench_extras._synthetic_states(rng, n, 10567)
rows = [int(i) for i in rng.integers(0, instances, size=n)]
inst = [i % instances for i in range(n)]
def step():
    fd = search.FlatDecoder(dec, store, ctx, mask)
    fd.seed(h_t, c_t)
    with torch.no_grad():
        fd.step(obs, udesc, rows, inst, k)
dt = bench_extras._timed(step, 2, 5)
rows_k, us = bench_extras.kernel_table(step, reps=3, top=30)
print('beam step over %d states: %.3f ms wall, %.3f ms of kernels' % (n, 1e3 * dt, 1e-3 * us))
for r in rows_k:
    print('   %-70s %6.1f calls %8.1f us avg %8.1f us/step %5.1f%%' % (r['kernel'][:70], r['calls_per_run'], r['avg_us'], r['us_per_run'], 100 * r['share']))
