"""Host / device breakdown of configs[4] on the FULL world: state_factored_search(K = 40, 1) over 64 instructions
(90 scans, 10 567 viewpoints) and the speaker rescoring of its candidates.

    python tools/search_profile.py [--cprofile]

    python tools/search_profile.py --numpy     # the round-3 form: numpy bookkeeping, host-issued decoder steps

Phases are wall-clock sections of the search with a device sync at each boundary (so the sum is a little above
the unsynchronised run, which is printed first): setup (env.reset, encoder, tables), per iteration the device step
(native path: one hipGraph replay + stream sync; numpy path: packing + one H2D + ~25 launches + one D2H), the
bookkeeping by difference (native: sim/frontier_core.cpp fill_inputs + advance; numpy: ~100 numpy calls), and
`results` (lineages, observation dictionaries, attention rows, physical walks).  Every timed search runs on a
minibatch the process has not seen before (fresh states: nothing is served from the env's sweep cache).
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
    wrap(frontier, '_setup_space', 'setup (env.reset, encoder pass, state space, graph inputs)')
    wrap(frontier, '_step_inputs', 'numpy path: inputs (gathers of the index arrays)', sync=False)
    wrap(search.FlatDecoder, 'step_arrays', 'numpy path: device step (pack + H2D + launches + top-k + D2H)')
    wrap(search.GraphStep, 'run', 'device step (hipGraph replay + stream sync)', sync=False)
    wrap(frontier, '_trajectories', 'results (lineages, observation dicts, attention rows)')
    wrap(frontier, 'physical_walks', 'results: physical walks', sync=False)

    def undo():
        for obj, name, fn in saved:
            setattr(obj, name, fn)
    return undo


def graph_replay_time(agent, reps=50):
    """Device time of one replay of the step graph (HIP events around back-to-back replays of the last inputs)."""
    gs = next(iter(agent._graph_steps.values()))
    keep = gs.pin_in.clone()
    gs.pin_in[7].fill_(-1)                          # (write no pool row)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gs.graph.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        gs.graph.replay()
    e1.record()
    torch.cuda.synchronize()
    gs.pin_in.copy_(keep)
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cprofile', action='store_true')
    ap.add_argument('--numpy', action='store_true', help='numpy bookkeeping + host-issued decoder steps')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    enc, dec, _, _ = bench.build_models(101, dev)
    enc.eval()
    dec.eval()
    table = bench.device_table(10567, 1234, dev)
    store = features.FeatureStore(table, device=dev)
    e64, _ = bench_extras.full_world(store, 64, seed=15, n_items=64 * 12)
    agent = agents.Seq2SeqAgent(e64, '/tmp/sf_search_profile.json', enc, dec, episode_len=8)
    agent.store = store
    if args.numpy:
        agent.search_backend = 'numpy'
    e64.set_beam_size(40)
    e64.reset_epoch()

    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            out = agent.state_factored_search(40, 1)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out
    for _ in range(2):
        run()
    ts = [run()[0] for _ in range(6)]
    print('state_factored_search(40, 1), 64 instructions, full world, %s path: %.1f ms mean / %.1f best / %.1f worst over 6 '
          'minibatches never seen before' % ('numpy' if args.numpy else 'native + graph', 1e3 * np.mean(ts), 1e3 * min(ts),
                                             1e3 * max(ts)))

    ph = Phases()
    undo = instrument(ph)
    dt, (trajs, completed, traversed) = run()
    undo()
    acc = sum(ph.t.values())
    print('\nwith a device sync at every phase boundary: %.1f ms, %d candidates' % (1e3 * dt, sum(len(x) for x in trajs)))
    print('%-62s %9s %7s %10s' % ('phase', 'ms', 'calls', 'us/call'))
    for k, v in sorted(ph.t.items(), key=lambda kv: -kv[1]):
        print('%-62s %9.2f %7d %10.1f' % (k, 1e3 * v, ph.n[k], 1e6 * v / ph.n[k]))
    print('%-62s %9.2f' % ('bookkeeping + the Python loop around it (by difference)', 1e3 * (dt - acc)))
    if not args.numpy:
        print('\none replay of the step graph (64 states: H2D, 2x nav look-up, gathers, decoder step, log-softmax, scatter, '
              'D2H): %.1f us on the device' % (1e6 * graph_replay_time(agent)))

    if args.cprofile:
        pr = cProfile.Profile()
        pr.enable()
        with torch.no_grad():
            agent.state_factored_search(40, 1)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(30)


if __name__ == '__main__':
    main()
