import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import test_gpu_hard_parity as T
from speaker_follower_amd import synth, features, follower as fol
def golden(name):
    with np.load(os.path.join(os.path.dirname(T.__file__), 'golden', name + '.npz')) as z:
        return {k: z[k] for k in z.files}
g = golden('g8_follower_peaky_b100_train')
for persistent in (True, False):
    enc, dec = T.follower(int(g['weight_seed'])); enc.train(); dec.train()
    enc.persistent = persistent
    fb = synth.follower_batch(seed=int(g['batch_seed']), batch=100, steps=20, n_viewpoints=256, stop_prob=1.0/40.0)
    store = features.FeatureStore(synth.feature_table(int(g['table_seed']), 256))
    batch = fol.DeviceFollowerBatch.from_synth(fb)
    eng = fol.FollowerEngine(enc, dec, store); eng.dropout_seed = int(g['dropout_seed'])
    st = eng.rollout(batch, 20, 'teacher', train=True)
    print('persistent', persistent, 'loss', float(st.loss), g['loss'])
    st.loss.backward()
    for pre, m in (('enc/', enc), ('dec/', dec)):
        for k, p in m.named_parameters():
            if p.grad is None or pre + 'gnorm/' + k not in g: continue
            n = float(p.grad.double().norm()); w = float(g[pre + 'gnorm/' + k])
            flat = p.grad.detach().cpu().numpy().ravel()
            idx = g[pre + 'gidx/' + k]; val = g[pre + 'gval/' + k]
            print('  %-50s norm %.6g want %.6g rel %.2e   sample max rel err %.2e' % (pre + k, n, w, abs(n - w) / max(w, 1e-30), np.max(np.abs(flat[idx] - val)) / (np.abs(val).max() + 1e-30)))
g = golden('g8_follower_peaky_b100_argmax')
enc, dec = T.follower(int(g['weight_seed'])); enc.eval(); dec.eval()
fb = synth.follower_batch(seed=int(g['batch_seed']), batch=100, steps=20, n_viewpoints=256)
store = features.FeatureStore(synth.feature_table(int(g['table_seed']), 256))
batch = fol.DeviceFollowerBatch.from_synth(fb)
with torch.no_grad():
    st = fol.FollowerEngine(enc, dec, store).rollout(batch, 20, 'argmax', train=False)
acts = st.actions.cpu().numpy()
bad = np.argwhere(acts != g['actions'])
print('action mismatches', len(bad), bad[:10].tolist())
got = st.logits.cpu().numpy(); want = g['logits']; fin = np.isfinite(want)
for t in range(20):
    d = np.abs(got[t][fin[t]] - want[t][fin[t]]).max()
    print('step %2d max logit err %.3e  (max |logit| %.2f)  alpha_v err %.2e' % (t, d, np.abs(want[t][fin[t]]).max(), np.abs(st.tape['alpha_v'][t].cpu().numpy() - g['alpha_v'][t]).max()))
for (t, b) in bad[:5]:
    print('mismatch at', t, b, 'got', acts[t, b], 'want', g['actions'][t, b], 'logits got', got[t, b][:fb.a_num[t, b]], 'want', want[t, b][:fb.a_num[t, b]])
