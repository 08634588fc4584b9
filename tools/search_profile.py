"""Host / device breakdown of configs[4] on the FULL world: state_factored_search(K = 40, 1) over 64 instructions
(90 scans, 10 567 viewpoints) and the speaker rescoring of its candidates.

    python tools/search_profile.py [--cprofile]

Phases are wall-clock sections of the search with a device sync at each boundary (so the sum is a little above
the unsynchronised run, which is printed first): setup (env.reset, encoder, tables), per iteration `inputs` (numpy
fancy indexing of the step's index arrays), `device step` (packing + one H2D + gathers + decoder step + top-k + one
D2H, and how much of that the device was busy: HIP events around the launches), `frontier` (numpy bookkeeping of the
successors), and `results` (lineages, observation dictionaries, attention rows).
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import torch                # noqa: E402

import bench                # noqa: E402
from speaker_follower_amd import bench_extras, features, frontier, search, agents    # noqa: E402


class Phases:
    def __init__(self):
        self.t = {}
        self.n = {}

    def add(self, name, dt):
        self.t[name] = self.t.get(name, 0.0) + dt
        self.n[name] = self.n.get(name, 0) + 1


def instrument(ph):
    """Wraps the search's building blocks with synchronised timers; returns the undo function."""
    saved = []

    def wrap(obj, name, label, sync=True):
        fn = getattr(obj, name)

        def timed(*a, **k):
            if sync:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*a, **k)
            if sync:
                torch.cuda.synchronize()
            ph.add(label, time.perf_counter() - t0)
            return out
        setattr(obj, name, timed)
        saved.append((obj, name, fn))
    wrap(frontier, '_setup', 'setup (env.reset, encoder pass, state space)')
    wrap(frontier, '_step_inputs', 'inputs (numpy gathers of the index arrays)', sync=False)
    wrap(search.FlatDecoder, 'step_arrays', 'device step (pack + H2D + decoder + top-k + D2H)')
    wrap(frontier, '_trajectories', 'results (lineages, observation dicts, attention rows)')
    wrap(frontier, 'physical_walks', 'results: physical walks', sync=False)

    def undo():
        for obj, name, fn in saved:
            setattr(obj, name, fn)
    return undo


def device_busy_of_step(agent, reps=20):
    """HIP-event time of ONE flat decoder step over 64 states (launches only, inputs already uploaded) next to its
    wall time: what the device is busy for inside `device step`."""
    env, space, fd, t, roots = frontier._setup(agent, True)
    inputs, _ = frontier._step_inputs(space, t, roots)
    fd.step_arrays(inputs, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fd.step_arrays(inputs, 0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    return wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cprofile', action='store_true')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    enc, dec, _, _ = bench.build_models(101, dev)
    enc.eval()
    dec.eval()
    table = bench.device_table(10567, 1234, dev)
    store = features.FeatureStore(table, device=dev)
    e64, _ = bench_extras.full_world(store, 64, seed=15)
    agent = agents.Seq2SeqAgent(e64, '/tmp/sf_search_profile.json', enc, dec, episode_len=8)
    agent.store = store
    e64.set_beam_size(40)

    def run():
        e64.reset_epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            out = agent.state_factored_search(40, 1)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out
    for _ in range(2):
        run()
    best = min(run()[0] for _ in range(5))
    print('state_factored_search(40, 1), 64 instructions, full world: %.1f ms (best of 5, unsynchronised phases)' % (1e3 * best))

    ph = Phases()
    undo = instrument(ph)
    dt, (trajs, completed, traversed) = run()
    undo()
    acc = sum(ph.t.values())
    print('\nwith a device sync at every phase boundary: %.1f ms, %d candidates' % (1e3 * dt, sum(len(x) for x in trajs)))
    print('%-62s %9s %7s %10s' % ('phase', 'ms', 'calls', 'us/call'))
    for k, v in sorted(ph.t.items(), key=lambda kv: -kv[1]):
        print('%-62s %9.2f %7d %10.1f' % (k, 1e3 * v, ph.n[k], 1e6 * v / ph.n[k]))
    print('%-62s %9.2f' % ('frontier bookkeeping + everything else (numpy, by difference)', 1e3 * (dt - acc)))
    e64.reset_epoch()
    wall = device_busy_of_step(agent)
    print('\none flat decoder step over 64 root states, back to back: %.1f us wall per step' % (1e6 * wall))

    if args.cprofile:
        pr = cProfile.Profile()
        e64.reset_epoch()
        pr.enable()
        with torch.no_grad():
            agent.state_factored_search(40, 1)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(30)


if __name__ == '__main__':
    main()
