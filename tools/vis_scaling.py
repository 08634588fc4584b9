"""(MODE=warm: the same rows in every call -- the upper bound of what prefetching the next step's rows could buy.)
How the visual attention (36 cold panorama rows per sample from the 3.1 GB table) scales with the number of
samples: per-workgroup-latency-bound (flat), per-CU-bandwidth-bound, or HBM-bound (linear)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import features, ops, model, synth, _lib
dev = torch.device('cuda', 0)
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
d = synth.FULL
rng = np.random.default_rng(0)
enc, dec, _, _ = bench.build_models(101, dev)
va = dec.visual_attention_layer
wv = (va.linear_in_h.weight, va.linear_in_h.bias, va.linear_in_v.weight, va.linear_in_v.bias)
MODE = os.environ.get('MODE', 'random')
for B in (16, 32, 64, 100, 128, 200, 256, 400):
    h = torch.randn(B, 512, device=dev)
    outs = []
    for rep in range(12):       # fresh viewpoints every call: cold rows (random, or one contiguous run of rows)
        if MODE == 'warm' and outs:     # the SAME rows every call: warm in the memory-side cache (what a prefetch could buy at best)
            outs.append(outs[0])
            continue
        if MODE in ('random', 'warm'):
            vp = torch.from_numpy(rng.integers(0, 10567, size=B).astype(np.int32)).to(dev)
        else:
            vp = torch.from_numpy(((rng.integers(0, 10567) + np.arange(B)) % 10567).astype(np.int32)).to(dev)
        view = torch.from_numpy(rng.integers(0, 36, size=B).astype(np.int32)).to(dev)
        outs.append((vp, view))
    torch.cuda.synchronize()
    with _lib.kernel_profile() as prof:
        for vp, view in outs:
            ops.visual_attention_fwd(wv, store.pano(vp, view), B, 36, d.feat, h)
    rows = {k: v for k, v in prof.rows.items() if 'visual_attn' in k}
    print('B %4d' % B, {k[:36]: round(v['avg_us'], 2) for k, v in rows.items()})
