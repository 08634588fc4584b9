"""Soak: many hipGraph replays of the headline rollout must reproduce the first one bit for bit, and a
few hundred training iterations must stay finite and reduce the loss (teacher forcing, fixed batch)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, dp, optim
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev); enc.eval(); dec.eval()
store = features.FeatureStore(bench.device_table(2048, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=2048)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
eng = follower.FollowerEngine(enc, dec, store)
replay, st = eng.capture(batch, 20, 'argmax')
replay(); torch.cuda.synchronize()
a0, l0, loss0 = st.actions.clone(), st.logits.clone(), st.loss_buf.clone()
N = int(os.environ.get('REPLAYS', 500))
bad = 0
for i in range(N):
    replay()
    if i % 50 == 49:
        torch.cuda.synchronize()
        same = torch.equal(st.actions, a0) and torch.equal(torch.nan_to_num(st.logits, neginf=0.), torch.nan_to_num(l0, neginf=0.)) and torch.equal(st.loss_buf, loss0)
        bad += 0 if same else 1
print('replays %d: %s' % (N, 'bit-identical' if bad == 0 else '%d mismatching checks' % bad))
# the state-factored search on its production path (one hipGraph replay + one native bookkeeping call per iteration, pools
# and graph kept on the agent): the same minibatch searched again and again must return the same candidates
from speaker_follower_amd import agents, bench_extras
M = int(os.environ.get('SEARCHES', 60))
store_full = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
e64, _ = bench_extras.full_world(store_full, 64, seed=15)
agent = agents.Seq2SeqAgent(e64, '/tmp/sf_soak_search.json', enc, dec, episode_len=8)
agent.store = store_full
e64.set_beam_size(40)
first, bads = None, 0
t0 = time.time()
with torch.no_grad():
    for i in range(M):
        e64.reset_epoch()
        trajs, _, walks = agent.state_factored_search(40, 1)
        rec = [[(c['actions'], c['score']) for c in cands] for cands in trajs], [w.nodes for w in walks]
        if first is None:
            first = rec
        bads += 0 if rec == first else 1
print('state-factored search x %d in %.2f s: %s' % (M, time.time() - t0, 'identical every time' if bads == 0 else '%d differ' % bads))
assert bads == 0
enc.train(); dec.train()
pe = [p for p in enc.parameters() if p.requires_grad]; pd = [p for p in dec.parameters() if p.requires_grad]
flat = dp.FlatGrads(pe + pd)
oe, od = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4), optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
eng2 = follower.FollowerEngine(enc, dec, store)
losses = []
T = int(os.environ.get('ITERS', 200))
for it in range(T):
    flat.zero()
    s2 = eng2.rollout(batch, 20, 'teacher', train=True)
    s2.loss.backward()
    oe.step(); od.step()
    if it % 20 == 0 or it == T - 1:
        losses.append(float(s2.loss.detach()))
import math
print('train %d iterations: losses %s finite=%s' % (T, ['%.3f' % v for v in losses], all(math.isfinite(v) for v in losses)))
assert bad == 0 and all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]
print('soak ok')
