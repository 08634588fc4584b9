"""Probe: follower rollout with the visual half of step t+1 on a side stream (device-flag ordering) vs the
paired single-stream schedule: equality of results and time per rollout (hipGraph replay)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
enc.eval(); dec.eval()
NV = int(os.environ.get('NV', 10567))
store = features.FeatureStore(bench.device_table(NV, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=NV)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
res = {}
for two in (False, True):
    eng = follower.FollowerEngine(enc, dec, store)
    eng.two_stream_forward = two
    with torch.no_grad():
        st = eng.rollout(batch, 20, 'argmax', train=False)
    torch.cuda.synchronize()
    replay, gst = eng.capture(batch, 20, 'argmax')
    for _ in range(5): replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    res[two] = (dt, gst.actions.clone(), gst.logits.clone(), float(gst.loss_buf))
    print('two_stream_forward=%s: %.3f ms per rollout (%.0f agent-steps/s), loss %.6f' % (two, dt * 1e3, 2000 / dt, res[two][3]))
print('actions equal', torch.equal(res[True][1], res[False][1]), 'max logit diff', float((res[True][2] - res[False][2]).nan_to_num(0, 0, 0).abs().max()))
from speaker_follower_amd import _lib
for two in (False, True):
    eng = follower.FollowerEngine(enc, dec, store)
    eng.two_stream_forward = two
    with torch.no_grad():
        eng.rollout(batch, 20, 'argmax', train=False)
        torch.cuda.synchronize()
        with _lib.kernel_profile() as prof:
            eng.rollout(batch, 20, 'argmax', train=False)
    tot = sum(r['total_us'] for r in prof.rows.values())
    print('two=%s kernel time sum %.0f us' % (two, tot))
    for k, r in sorted(prof.rows.items(), key=lambda kv: -kv[1]['total_us'])[:16]:
        print('   %-56s calls %3d avg %7.2f total %7.1f' % (k[:56], r['calls'], r['avg_us'], r['total_us']))
