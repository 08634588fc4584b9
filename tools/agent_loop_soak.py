"""Soak of the agents' loops as train.py drives them (train.py:97-111): rounds of Seq2SeqAgent.train (whole iterations as
graph replays, plain torch.optim.Adam adopted) followed by Seq2SeqAgent.test (one inference graph replay per minibatch)
on the full world.  After every round the graph-replayed test must give the results of the launch-by-launch rollout
with the SAME weights (a captured inference graph reads derived weight copies: they must follow the optimizer steps);
no fallbacks, no growth of device memory, a bounded number of captured graphs."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import bench_extras, features, agents
dev = torch.device('cuda', 0)
enc, dec, _, _ = bench.build_models(101, dev)
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
e_train, nt = bench_extras.full_world(store, 100, seed=41, n_items=1500)
ag = agents.Seq2SeqAgent(e_train, '/tmp/agent_soak.json', enc, dec, episode_len=20)
ag.store = store
ag.use_device_env(nt)
oe = torch.optim.Adam([p for p in enc.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
od = torch.optim.Adam([p for p in dec.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
ROUNDS = int(os.environ.get('ROUNDS', 12))
mem = []
t0 = time.perf_counter()
for r in range(ROUNDS):
    ag.train(oe, od, 6, feedback='sample')
    assert all(l == l and abs(l) < 1e6 for l in ag.losses), ag.losses
    ag.test_graph = True
    a = ag.test(use_dropout=False, feedback='argmax')
    ag.test_graph = False
    b = ag.test(use_dropout=False, feedback='argmax')
    ag.test_graph = True
    assert sorted(a) == sorted(b) and len(a) == 1500
    diff = sum(1 for k in a if a[k]['actions'] != b[k]['actions'])
    worst = max(abs(a[k]['score'] - b[k]['score']) for k in a)
    gc.collect()                                        # (the launch-by-launch rollouts leave their states to the collector)
    mem.append(torch.cuda.memory_allocated(dev) >> 20)
    print('round %2d: train loss %.3f  test: %d / %d action sequences differ, max |score difference| %.2e, %d MB allocated, '
          '%d inference graphs, fallbacks %d' % (r, ag.losses[-1], diff, len(a), worst, mem[-1], len(ag._test_graphs),
                                                 ag._engine.fallbacks), flush=True)
    # (fixed-width padding changes the summation order of the padded attention columns: a 0-ulp tie may flip a walk)
    assert diff <= 2 and ag._engine.fallbacks == 0 and len(ag._test_graphs) == 1
assert max(mem[2:]) - min(mem[2:]) <= 64, mem
print('agent loop soak ok: %d rounds in %.1f s' % (ROUNDS, time.perf_counter() - t0))
