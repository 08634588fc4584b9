import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch, time
import test_gpu_mega as T
from speaker_follower_amd import _lib
from speaker_follower_amd.model import decoder_params
from speaker_follower_amd.runtime import ptr, ws_args, workspace
S = int(os.environ.get('S', 1))
for B in [int(b) for b in os.environ.get('BS', '16,20,32,48,100').split(',')]:
    enc, dec, store, batch, st = T.reference_rollout(B, S)
    p = decoder_params(dec)
    lw = _lib.LstmW(p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(), p[3].data_ptr(), None, None)
    ref = st.tape['h1'].cpu().numpy(); refc = st.tape['c1'].cpu().numpy(); refg = st.tape['gates'].cpu().numpy()
    h1 = torch.full((S, B, 512), float('nan'), device='cuda'); c1 = torch.full((S, B, 512), float('nan'), device='cuda')
    g = torch.full((S, B, 2048), float('nan'), device='cuda')
    t0 = time.time()
    _lib.call('sf_debug_mega_lstm_loop', C.byref(lw), ptr(st.h_init), ptr(st.c_init), ptr(st.tape['xin']), B, S, ptr(h1), ptr(c1), ptr(g), *ws_args(h1.device))
    torch.cuda.synchronize(); dt = time.time() - t0
    x = workspace(h1.device)[:3 * 128 * 4864 * 4].view(torch.float32).view(3, 128, 4864).cpu().numpy()
    print('B', B, 'time %.3f' % dt)
    for name, got, want in (('h1', h1, ref), ('c1', c1, refc), ('gates', g, refg)):
        got = got.cpu().numpy()
        for t in range(S):
            out = []
            for r in range(0, B, 16):
                a, w = got[t, r:r + 16], want[t, r:r + 16]
                n = np.isnan(a)
                out.append('%.2f/%.0e' % (n.mean(), np.abs(a - w)[~n].max() if (~n).any() else -1))
            print('  ', name, 't', t, 'per group nanfrac/err:', ' '.join(out))
    for t in range(S):
        xh = x[(t + 1) % 3, :B, 4352:]
        out = []
        for r in range(0, B, 16):
            a, w = xh[r:r + 16], ref[t, r:r + 16]
            n = np.isnan(a)
            out.append('%.2f/%.0e' % (n.mean(), np.abs(a - w)[~n].max() if (~n).any() else -1))
        print('   XIN h after t', t, ' '.join(out))
