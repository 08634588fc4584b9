"""Probe: how fast would the training iteration be as ONE replayed hipGraph?  (Captured with fixed
dropout sites and a fixed Adam step count, so the replays are not a valid training run -- timing only.)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, dp, optim
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev); enc.train(); dec.train()
store = features.FeatureStore(bench.device_table(2048, 1234, dev), device=dev)
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=2048)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
pe = [p for p in enc.parameters() if p.requires_grad]; pd = [p for p in dec.parameters() if p.requires_grad]
flat = dp.FlatGrads(pe + pd)
oe, od = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4), optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
eng = follower.FollowerEngine(enc, dec, store)
eng.two_stream_backward = os.environ.get('TWO', '0') == '1'
def it():
    flat.zero()
    st = eng.rollout(batch, 20, 'argmax', train=True)
    st.loss.backward()
    oe.step(); od.step()
for _ in range(5): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): it()
torch.cuda.synchronize(); print('eager   %.3f ms / iteration' % ((time.perf_counter() - t0) / 20 * 1e3))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
from speaker_follower_amd import runtime
runtime.ensure_workspace(s, dev)          # (a workspace first touched inside a capture would be zero-filled by every replay)
with torch.cuda.stream(s):
    it()                                   # warm-up on the capture stream (side streams, caches)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        it()
torch.cuda.current_stream().wait_stream(s)
for _ in range(5): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); print('replay  %.3f ms / iteration' % ((time.perf_counter() - t0) / 20 * 1e3))
