"""A/B of the inference decode chain on the headline batch (100 x 20 steps, hipGraph replay, ms per rollout):
unfolded (round 5's six launches behind the cell), folded text attention with four launches (partials + ticket merge in the r
launch | partials beside r, merge beside scoring + glue), and the three-launch chain over the M_v / M_a products."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, _lib
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
enc.eval(); dec.eval()
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=10567)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)


def timed(replay, n=40):
    for _ in range(8):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


ref = None
for name, fold, chain, merge in (('unfolded (six launches behind the cell)', False, False, 1),
                                 ('folded, partials + ticket merge beside r', True, False, 0),
                                 ('folded, partials beside r, merge beside the glue', True, False, 1),
                                 ('folded, three launches (M_v / M_a products)', True, True, 1)):
    _lib.lib.sf_debug_fold_merge_with_glue(merge)
    eng = follower.FollowerEngine(enc, dec, store)
    eng.fold_text, eng.fold_chain = fold, chain
    replay, st = eng.capture(batch, 20, 'argmax')
    ms = [timed(replay) for _ in range(3)]
    acts = st.actions.clone()
    if ref is None:
        ref = acts
    print('%-52s %.4f ms per rollout (best of 3; %.4f .. %.4f) = %7.0f agent-steps/s, actions equal: %s'
          % (name, min(ms), min(ms), max(ms), 2000 / (min(ms) * 1e-3), bool(torch.equal(acts, ref))), flush=True)
_lib.lib.sf_debug_fold_merge_with_glue(1)
