import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from speaker_follower_amd import synth, model, features, follower
d = synth.FULL
enc_w, dec_w = synth.follower_weights(0)
enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()}); dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
enc.cuda().eval(); dec.cuda().eval()
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=512, min_len=10, max_len=80, a_max=14)
store = features.FeatureStore(synth.feature_table(0, 512))
batch = follower.DeviceFollowerBatch.from_synth(fb)
res = {}
for fold in (False, True):
    eng = follower.FollowerEngine(enc, dec, store); eng.fold_inference = fold
    replay, st = eng.capture(batch, 20, 'argmax')
    for _ in range(5): replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    res[fold] = (st.actions.clone(), st.logits.clone())
    print('fold', fold, '%.3f ms' % (dt * 1e3), '%.0f agent-steps/s' % (2000 / dt))
print('actions equal', torch.equal(res[False][0], res[True][0]), 'max logit diff', (res[False][1] - res[True][1]).abs()[torch.isfinite(res[False][1])].max().item())
