"""Inner timeline of the persistent speaker word loop (sf_debug_trace) + kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from speaker_follower_amd import _lib, speaker
import test_gpu_persistent as tp
enc, dec, store, batch = tp.speaker_setup(100)
S = 80
for fb in ('argmax', 'teacher'):
    eng = speaker.SpeakerEngine(enc, dec, store)
    with torch.no_grad():
        eng.score(batch, S, fb, train=False)
        with _lib.kernel_profile() as prof:
            eng.score(batch, S, fb, train=False)
        print(fb, {k: round(v['total_us'], 1) for k, v in sorted(prof.rows.items(), key=lambda kv: -kv[1]['total_us'])[:6]})
        trace = torch.zeros(256 * 8, dtype=torch.int64, device='cuda')
        _lib.lib.sf_debug_trace(trace.data_ptr())
        eng.score(batch, S, fb, train=False)
        torch.cuda.synchronize()
        _lib.lib.sf_debug_trace(None)
    t = trace.cpu().numpy().reshape(256, 8).astype(np.float64) / 100.0 / S
    for k, n in enumerate(['cell + partial scores + publish', 'wait h1 + scores', 'softmax + MFMA x5 + reduce', 'h~ + publish',
                           'wait h~', 'vocab MFMA + stats + publish', 'wait stats + combine', 'outputs + table row + loop']):
        print('    %-34s mean %.2f  min %.2f  max %.2f us/step' % (n, t[:, k].mean(), t[:, k].min(), t[:, k].max()))
