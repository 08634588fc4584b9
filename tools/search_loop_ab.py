"""A/B of the state-factored search's iteration loop on one box: the native loop (sim/frontier_core.cpp: run_graph --
inputs, graph launch, stream sync and bookkeeping without returning to Python) against the Python loop, alternating, on
minibatches never seen before (full world, K = 40, 64 instructions)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from speaker_follower_amd import bench_extras, features, agents
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
enc.eval(); dec.eval()
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
e64, _ = bench_extras.full_world(store, 64, seed=15, n_items=64 * 30)
agent = agents.Seq2SeqAgent(e64, '/tmp/sf_search_ab.json', enc, dec, episode_len=8)
agent.store = store
e64.set_beam_size(40)
e64.reset_epoch()
def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        agent.state_factored_search(40, 1)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for _ in range(3): run()
ts = {True: [], False: []}
for i in range(24):
    agent.search_native_loop = (i % 2 == 0)
    ts[agent.search_native_loop].append(run())
for k in (True, False):
    v = np.array(ts[k]) * 1e3
    print('%-12s mean %.2f ms  median %.2f  best %.2f  worst %.2f  (%d minibatches)' % ('native loop' if k else 'python loop', v.mean(), np.median(v), v.min(), v.max(), len(v)))
