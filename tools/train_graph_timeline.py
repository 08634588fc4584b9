"""Driver for a kernel timeline of the follower's training iteration as hipGraph replays (bench `train_iteration`):
rocprofv3 --kernel-trace -- python3 this; tools/prof_train_timeline.sh graph prints the last replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, dp, optim
dev = torch.device('cuda', 0)
NV = int(os.environ.get('NV', 10567))
store = features.FeatureStore(bench.device_table(NV, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=NV)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
enc, dec, _, _ = bench.build_models(101, dev)
enc.train(); dec.train()
flat = dp.BucketedGrads(dp.follower_buckets(enc, dec), group=None)
oe = optim.FusedAdam([p for p in enc.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
od = optim.FusedAdam([p for p in dec.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
eng = follower.FollowerEngine(enc, dec, store)
from speaker_follower_amd import _lib
if 'SF_BPTT_FLAGS' in os.environ:
    _lib.lib.sf_debug_bptt_flags(int(os.environ['SF_BPTT_FLAGS']))
if 'SF_LOOKAHEAD' in os.environ:
    _lib.lib.sf_debug_bptt_lookahead(int(os.environ['SF_LOOKAHEAD']))
if 'SF_FUSED_CELL' in os.environ:
    _lib.lib.sf_debug_fused_cell_backward(int(os.environ['SF_FUSED_CELL']))
if 'SF_SLAB_CONSUMERS' in os.environ:
    _lib.lib.sf_debug_slab_consumers(int(os.environ['SF_SLAB_CONSUMERS']))
if 'SF_GROUPED' in os.environ:
    _lib.lib.sf_debug_grouped_weight_gradients(int(os.environ['SF_GROUPED']))
for k, v in os.environ.items():                       # e.g. SF_ENGINE_split_wgrad_streams=1
    if k.startswith('SF_ENGINE_'):
        setattr(eng, k[len('SF_ENGINE_'):], int(v))
for _ in range(3):
    flat.zero(); st = eng.rollout(batch, 20, 'argmax', train=True); st.loss.backward(); oe.step(); od.step()
tg = eng.capture_training(batch, 20, 'argmax', optimizers=(oe, od), zero=flat)
import time
for _ in range(3):
    tg.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(int(os.environ.get('REPS', 6))):
    tg.replay()
torch.cuda.synchronize()
print('graph replay %.3f ms per iteration' % ((time.perf_counter() - t0) / int(os.environ.get('REPS', 6)) * 1e3))
