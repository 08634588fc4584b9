"""Seq2SeqSpeaker.train (speaker.py:376-395, train_speaker.py:28-31) through the agents' API on the full world, a new
minibatch of 100 every iteration: the minibatch's gold routes from the navigation tables (nav.NavTable.gold_routes, the
default) against the lock-step walk of the host environment (index_gold_routes = False), first epoch (states never seen:
the walk sweeps the simulator) and second epoch (its sweep cache is warm)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['bench.py']
import bench
from speaker_follower_amd import bench_extras, features, agents, optim
dev = torch.device('cuda', 0)
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
for index in (True, False):
    senc, sdec = bench_extras._speaker_models(dev)
    e, _ = bench_extras.full_world(store, 100, seed=31 + index, n_items=1000)
    spk = agents.Seq2SeqSpeaker(e, '/tmp/x.json', senc, sdec, 80)
    spk.store = store
    spk.index_gold_routes = index
    oe = optim.FusedAdam([p for p in senc.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
    od = optim.FusedAdam([p for p in sdec.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
    e.reset_epoch()
    spk.train(oe, od, 2, feedback='teacher')                 # (caches, workspaces)
    for epoch in ('first epoch ', 'second epoch'):
        if epoch == 'second epoch':
            e.reset_epoch()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        spk.train(oe, od, 8, feedback='teacher')
        torch.cuda.synchronize()
        print('%-28s %s: %.2f ms per iteration' % ('routes from the tables' if index else 'walk of the host environment', epoch,
                                                   (time.perf_counter() - t0) / 8 * 1e3))
