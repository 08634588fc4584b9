"""Extra measurements of bench.py (never its `value`): the other BASELINE.json configs on one GPU.

* `speaker_decode`  -- configs[2]: SpeakerEncoderLSTM over 100 paths of 4-7 steps + 80 greedy
  SpeakerDecoderLSTM word steps (data_augmentation_from_speaker.py -> speaker.py:158-197), and the
  teacher-forced scoring of the same batch (what pragmatic re-ranking runs per candidate).
* `search_step`     -- configs[4]: one expansion step of the follower search over a flat list of
  states (follower.py:575-603 / 785-803: gather h/c rows, one AttnDecoderLSTM step with per-state
  instruction rows, log-softmax + sorted top-k) at 64 states (state-factored search, one state per
  instance) and at 64 x 40 states (beam form), plus speaker scoring of 64 x 40 candidate paths
  (rational_follower.py:60-95).

Every measurement is wrapped: a failure is reported as {"error": ...} instead of losing the line.
"""
import os
import time
import traceback

import numpy as np
import torch


def _guard(fn):
    def run(*a, **k):
        try:
            return fn(*a, **k)
        except Exception as e:                                    # noqa: BLE001
            return dict(error='%s: %s' % (type(e).__name__, e), where=traceback.format_exc(limit=2))
    return run


def _timed(fn, warm, reps):
    import gc
    was = gc.isenabled()
    gc.collect()                # (before the warm-up: the device clocks fall during host-only time)
    gc.disable()                # a full collection inside ten 2 ms replays once turned 1.9 ms into 8.9 (round 5)
    try:
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    finally:
        if was:
            gc.enable()


def kernel_table(fn, reps=2, top=16):
    """Per-kernel time of `fn` measured IN THIS RUN: HIP start / stop events on every dispatch the library makes
    (sf_profile_begin / sf_profile_end -- the kernel's own execution time on its launch stream, what rocprofv3
    --kernel-trace reports).  Eager issue only.  Returns (rows, kernel-us per call of fn)."""
    from . import _lib
    fn()
    torch.cuda.synchronize()
    with _lib.kernel_profile() as prof:
        for _ in range(reps):
            fn()
    total = sum(r['total_us'] for r in prof.rows.values())
    rows = [dict(kernel=k, calls_per_run=r['calls'] / reps, avg_us=r['avg_us'], us_per_run=r['total_us'] / reps,
                 share=r['total_us'] / total)
            for k, r in sorted(prof.rows.items(), key=lambda kv: -kv[1]['total_us'])[:top]]
    return rows, total / reps


def _speaker_models(device, seed=5):
    from . import synth, model
    d = synth.FULL
    enc_w, dec_w = synth.speaker_weights(seed)
    enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
    dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=dec_w['embedding.weight'])
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return enc.to(device).eval(), dec.to(device).eval()


@_guard
def speaker_decode(store, device, batch=100, words=80):
    from . import synth, speaker
    enc, dec = _speaker_models(device)
    n_vp = store.table.shape[0]
    sb = synth.speaker_batch(seed=0, batch=batch, n_viewpoints=n_vp, min_path=4, max_path=7,
                             min_len=10, max_len=79)
    b = speaker.DeviceSpeakerBatch.from_synth(sb, device=device)
    eng = speaker.SpeakerEngine(enc, dec, store)
    out = dict(what='speaker: %d paths of 4-7 steps (encoder: visual attention + LSTMCell per path step) + %d word '
                    'steps, hipGraph replay' % (batch, words), unit='word-steps/s')
    for fb in ('argmax', 'teacher'):
        replay, _ = eng.capture(b, words, fb)
        dt = _timed(replay, 3, 10)
        out['greedy_decode' if fb == 'argmax' else 'teacher_scoring'] = dict(
            value=batch * words / dt, ms_per_batch=1e3 * dt)

    def eager():
        with torch.no_grad():
            eng.score(b, words, 'argmax', train=False)
    rows, us = kernel_table(eager)
    out['kernels'] = rows
    out['kernel_time_ms_per_batch'] = 1e-3 * us
    # what the path executes: encoder Tp x (visual attention + gate product [B,2F+H] x [4H,2F+H]^T), decoder
    # `words` x (recurrent product, attention over <= 7 path steps, h~, vocabulary projection)
    d = synth.FULL
    H, F, V = d.hidden, d.feat, d.vocab
    Tp = int(sb.vp.shape[0])
    flops = batch * (Tp * 2.0 * ((2 * F + H) * 4 * H + H * d.dot + d.dot * F + 2 * 36 * F)
                     + words * 2.0 * (H * 4 * H + H * H + 2 * Tp * H + 2 * H * H + H * V))
    ms = out['greedy_decode']['ms_per_batch']
    out['roofline'] = dict(executed_gflop_per_batch=flops / 1e9, tflops=flops / (ms * 1e-3) / 1e12,
                           mfma_frac=flops / (ms * 1e-3) / 1e12 / 157.3,
                           note='executed FLOPs of one batch / hipGraph-replay time; the decoder is %d dependent word '
                                'steps in ONE persistent launch (spk_persist_kernel): latency-, not roofline-bound' % words)
    return out


@_guard
def speaker_sweep(store, device, n_paths=178300, batch=100, words=80):
    """configs[2] as the reference runs it (data_augmentation_from_speaker.py:56-58: Seq2SeqSpeaker.test with argmax
    feedback over the 178 300 sampled trajectories, one minibatch after the other), MEASURED over all `n_paths`
    distinct synthetic paths (4-7 steps, ragged inside every minibatch) in minibatches of `batch`: host packing of
    every index batch into pinned memory, one H2D copy, greedy decoding of `words` words, and the D2H copy of the
    generated word ids are all inside the timed region (speaker.SpeakerSweep: two streams, one hipGraph per stream
    and path-step count, packing of minibatch n+1 beside the device work of n).  The data set itself -- the index
    arrays of the paths -- is generated before the clock starts, as the reference loads its json before its loop."""
    from . import synth, speaker
    import time
    enc, dec = _speaker_models(device)
    n_vp = store.table.shape[0]
    nb = (n_paths + batch - 1) // batch
    sbs = [synth.speaker_batch(seed=500 + i, batch=batch, n_viewpoints=n_vp, min_path=4, max_path=7, min_len=10,
                               max_len=79) for i in range(nb)]
    sweep = speaker.SpeakerSweep(enc, dec, store, batch, words)
    sweep.run(sbs[:40])                                  # warm-up: graph capture for every path-step count, clocks
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sweep.run(sbs)
    dt = time.perf_counter() - t0
    # the same minibatches one after the other on ONE stream, eager issue (round 3's way), on a sample
    eng = speaker.SpeakerEngine(enc, dec, store)

    def eager():
        res = []
        with torch.no_grad():
            for sb in sbs[:20]:
                st = eng.score(speaker.DeviceSpeakerBatch.from_synth(sb, device=device), words, 'argmax', train=False)
                res.append(st.words[1:].cpu())
        return res
    dt_e = _timed(eager, 1, 2)
    ref = torch.stack(eager()).numpy()
    n = nb * batch
    # The decoded words do not depend on how the paths are grouped: with the persistent word loop's full 128 rows per
    # launch (8 row groups x 16) the same data set needs 22 % fewer minibatches of the same latency
    wide = None
    if batch == 100:
        nb2 = (n_paths + 127) // 128
        sbs2 = [synth.speaker_batch(seed=90500 + i, batch=128, n_viewpoints=n_vp, min_path=4, max_path=7, min_len=10,
                                    max_len=79) for i in range(nb2)]
        sweep2 = speaker.SpeakerSweep(enc, dec, store, 128, words)
        sweep2.run(sbs2[:40])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sweep2.run(sbs2)
        dt2 = time.perf_counter() - t0
        wide = dict(value=nb2 * 128 / dt2, unit='paths/s', seconds=dt2, paths=nb2 * 128, minibatch=128,
                    ms_per_minibatch=1e3 * dt2 / nb2,
                    note='the same sweep in minibatches of 128 (a free choice for inference: greedy decoding is per path)')
    return dict(minibatch_128=wide, what='greedy speaker decoding of %d distinct paths (4-7 steps) x %d words in minibatches of %d: host packing '
                     '+ one H2D copy + decode (hipGraph per stream and path-step count, two streams) + D2H of the words per '
                     'minibatch, ALL inside the timed region' % (n, words, batch),
                value=n / dt, unit='paths/s', seconds=dt, paths=n, ms_per_minibatch=1e3 * dt / nb,
                host_packing_seconds=sweep.host_pack_s,
                words_equal_single_stream_eager=bool((out[:20].astype('int64') == ref).all()),
                single_stream_eager_ms_per_minibatch=1e3 * dt_e / 20,
                note='measured over the whole sweep, nothing extrapolated')


@_guard
def speaker_parity_g9(device):
    """The speaker's HARD parity set (golden G9: B = 100, peaky weights, attention scores up to +-80, |logit| up to 17),
    measured in this run: max |logit difference| of the HIP path at the first and the last word step against (a) the
    REFERENCE's own fp32 output (north_star's letter: 1e-4) and (b) the same reference modules evaluated in float64.
    The reference's fp32 run is itself 1e-4 .. 2.3e-4 from its float64 evaluation; the HIP path (float64 attention
    query / scores in the path encoder, bf16x6 products) sits ~3e-5 from float64 -- so its distance to the fp32
    reference is dominated by the REFERENCE's rounding and exceeds 1e-4 on this set: a known deviation from the
    letter, stated here instead of hidden behind the float64 assertion.  Words are bit-exact."""
    import os
    from . import synth, model, features, speaker
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        with np.load(os.path.join(root, 'tests', 'golden', name + '.npz')) as z:
            return {k: z[k] for k in z.files}
    f64 = load('g9_speaker_b100_f64')
    out = {}
    for feedback in ('argmax', 'teacher'):
        g = load('g9_speaker_b100_' + feedback)
        d = synth.FULL
        senc_w, sdec_w = synth.speaker_weights_peaky(int(g['weight_seed']))
        enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
        dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=sdec_w['embedding.weight'])
        enc.load_state_dict({k: torch.tensor(v) for k, v in senc_w.items()})
        dec.load_state_dict({k: torch.tensor(v) for k, v in sdec_w.items()})
        enc.to(device).eval()
        dec.to(device).eval()
        sb = synth.speaker_batch(seed=int(g['batch_seed']), batch=100, n_viewpoints=256, min_len=10, max_len=79)
        store = features.FeatureStore(synth.feature_table(int(g['table_seed']), 256), device=device)
        n = int(g['n_steps'])
        with torch.no_grad():
            st = speaker.SpeakerEngine(enc, dec, store).score(speaker.DeviceSpeakerBatch.from_synth(sb, device=device), n,
                                                              feedback, train=False)
        lg = st.logits.cpu().numpy()
        V = g['logit_last'].shape[-1]
        first, last = lg[0][:, :V], lg[n - 1][:, :V]
        out[feedback] = dict(
            words_bit_exact=bool(np.array_equal(st.words[1:].cpu().numpy(), g['words'])), word_steps=n,
            max_abs_logit=float(np.abs(g['logit_last']).max()),
            vs_fp32_reference=float(max(np.abs(first - g['logits_first'][0]).max(), np.abs(last - g['logit_last']).max())),
            vs_float64=float(max(np.abs(first - f64[feedback + '/logits_first']).max(),
                                 np.abs(last - f64[feedback + '/logit_last']).max())),
            reference_fp32_vs_its_float64=float(max(f64[feedback + '/ref32_dist_first'], f64[feedback + '/ref32_dist_last'])))
    out['note'] = ('north_star asks for 1e-4 of the reference CPU path: met against the float64 evaluation of the reference '
                   'modules; vs the reference\'s OWN fp32 output this set measures ~2e-4 because that output is itself '
                   '1e-4 .. 2.3e-4 from exact arithmetic (tests/test_gpu_hard_parity.py)')
    return out


@_guard
def search_full(conn_dir, device, scans=('YmJkqBEsHnH', 'gZ6f7yhEvPG', 'GdvgFV5R1Z5'), instances=64, k=40,
                episode_len=8):
    """configs[4] end to end on real connectivity graphs: Seq2SeqAgent.state_factored_search(K = 40, 1)
    over one minibatch of 64 instructions (follower.py:720-980), host bookkeeping and the navigation-only
    simulator included.  The same search is pinned against the reference's own output in
    tests/test_gpu_search.py (there the reference took 17.8 s on 4 CPU threads)."""
    import os
    from . import env, synth, model, features, agents
    from .build import build_sim
    build_sim(verbose=False)
    graphs = {s: env.NavGraph(os.path.join(conn_dir, s + '_connectivity.json')) for s in scans}
    items = env.random_items(graphs, instances, np.random.default_rng(15), min_len=4, max_len=20)
    row_of, n = {}, 0
    for s, g in graphs.items():
        for v in g.ids:
            row_of[s + '_' + v] = n
            n += 1
    table = synth.feature_table(11, n)
    e = env.R2RIndexEnv(items, row_of, conn_dir, batch_size=instances, host_table=None)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(303)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({kk: torch.tensor(v) for kk, v in enc_w.items()})
    dec.load_state_dict({kk: torch.tensor(v) for kk, v in dec_w.items()})
    enc.to(device).eval()
    dec.to(device).eval()
    agent = agents.Seq2SeqAgent(e, '/tmp/sf_bench_search.json', enc, dec, episode_len=episode_len)
    agent.store = features.FeatureStore(table, device=device)
    e.set_beam_size(k)
    best, n_c, n_steps = None, 0, 0
    for _ in range(3):
        e.reset_epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            trajs, completed, traversed = agent.state_factored_search(k, 1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        n_c = sum(len(t) for t in trajs)
    # follower.py:541-718 on the same minibatch: plain beam search, K hypotheses per instruction
    best_b = None
    for _ in range(3):
        e.reset_epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            agent.beam_search(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best_b = dt if best_b is None else min(best_b, dt)
    return dict(what='state_factored_search(K=%d, 1), %d instructions, %d-viewpoint fixture graphs, episode_len %d'
                     % (k, instances, n, episode_len), value=instances / best, unit='instructions/s',
                seconds=best, candidates=n_c,
                beam_search=dict(seconds=best_b, value=instances / best_b, unit='instructions/s',
                                 what='beam_search(K=%d) over the same minibatch (up to %d states per step)' % (k, instances * k)))


@_guard
def real_env_rollout(conn_dir, device, batch=100, steps=20, scans=('YmJkqBEsHnH', 'gZ6f7yhEvPG', 'GdvgFV5R1Z5')):
    """configs[1] inference half on REAL connectivity graphs: a student-forced (argmax) follower rollout
    whose every next observation depends on the action just chosen -- env.step / observe / teacher run
    on the device (nav.py, sf_nav_step), so head(t+1) cannot be pipelined beside tail(t) -- next to the
    SAME rollout driven the reference's way (agents._rollout_with_loss: D2H of the actions and Python
    env.step / observe every step, follower.py:507-514)."""
    import os
    from . import env, synth, model, features, agents, nav, follower
    from .build import build_sim
    build_sim(verbose=False)
    graphs = {s: env.NavGraph(os.path.join(conn_dir, s + '_connectivity.json')) for s in scans}
    items = env.random_items(graphs, batch, np.random.default_rng(21), min_len=10, max_len=79)
    row_of, n = {}, 0
    for s, g in graphs.items():
        for v in g.ids:
            row_of[s + '_' + v] = n
            n += 1
    table = synth.feature_table(11, n)
    e = env.R2RIndexEnv(items, row_of, conn_dir, batch_size=batch, host_table=table)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(303)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({kk: torch.tensor(v) for kk, v in enc_w.items()})
    dec.load_state_dict({kk: torch.tensor(v) for kk, v in dec_w.items()})
    enc.to(device).eval()
    dec.to(device).eval()
    store = features.FeatureStore(table, device=device)
    nt = nav.NavTable(e, store)
    e.reset_epoch()
    e._next_minibatch(True)
    navb = nav.DeviceNavBatch(nt, list(e.batch), steps)
    eng = follower.FollowerEngine(enc, dec, store)
    replay, gst = eng.capture(navb, steps, 'argmax')
    dt = _timed(replay, 3, 10)
    out = dict(what='CACHE-RESIDENT TABLE (%d viewpoints = %.0f MB: every panorama is served by L2 / MALL; the full-size '
                    'number is real_env_full): student-forced argmax rollout on three fixture graphs, batch %d, %d '
                    'decode steps, encoder included; every step executed for every row'
                    % (n, n * 36 * 2048 * 4 / 1e6, batch, steps),
               unit='agent-steps/s', device_env=dict(value=batch * steps / dt, ms_per_rollout=1e3 * dt,
                                                     launch='hipGraph replay, one host sync per rollout'))
    agent = agents.Seq2SeqAgent(e, '/tmp/sf_bench_nav.json', enc, dec, episode_len=steps)
    agent.feedback = 'argmax'

    def host():
        e.reset_epoch()
        with torch.no_grad():
            agent._rollout_with_loss()
    host()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host()
    torch.cuda.synchronize()
    dth = time.perf_counter() - t0
    out['host_env_per_step'] = dict(ms_per_rollout=1e3 * dth,
                                    note='reference-style loop: dense observations, D2H + Python env every step; '
                                         'exits early once every row has stopped')
    return out


_FULL_WORLD = {}


def full_world(store, batch=100, seed=21, n_items=None):
    """The FULL-SIZE real environment of BASELINE configs[1]: all 90 connectivity graphs (10 567 included
    viewpoints, data/r2r_connectivity.npz) over a feature table with one row per viewpoint, a NavTable of
    every (viewpoint, view) state, and `n_items` (default: one minibatch) random items (2-5 hop shortest paths,
    10-79 token instructions) served in minibatches of `batch`.  Cached per store: the host-side build takes a
    few seconds."""
    from . import env, nav, nav_data
    from .build import build_sim
    key = (id(store), batch, seed, n_items)
    if key not in _FULL_WORLD:
        build_sim(verbose=False)
        geo = nav_data.load_geometry()
        row_of, n = nav_data.row_index(geo)
        if store.n != n:
            raise RuntimeError('the full-size environment needs a %d-row feature table (got %d)' % (n, store.n))
        conn = nav_data.connectivity_dir()
        shared = _FULL_WORLD.get(('nav', id(store)))               # the parsed graphs and the state tables: once per store
        graphs = shared[0] if shared else {s: env.NavGraph(os.path.join(conn, s + '_connectivity.json')) for s in geo}
        items = env.random_items(graphs, n_items or batch, np.random.default_rng(seed), min_len=10, max_len=79)
        e = env.R2RIndexEnv(items, row_of, conn, batch_size=batch)
        e.graphs = graphs                       # all 90 (already parsed: the objects the items were drawn from)
        nt = shared[1] if shared else nav.NavTable(e, store)
        _FULL_WORLD[('nav', id(store))] = (graphs, nt)
        e._nav_table = (store, nt, tuple(sorted(e.graphs)))        # what nav.table_for(e, store) hands the search
        _FULL_WORLD[key] = (e, nt)
        # The world is ~10^6 long-lived Python objects (90 parsed graphs, the items): moved out of the cyclic
        # collector's sight, or every later full collection walks them -- milliseconds added to whatever host-bound
        # measurement it happens to interrupt (what a long-running training / evaluation process does once after set-up)
        import gc
        gc.collect()
        gc.freeze()
    return _FULL_WORLD[key]


@_guard
def real_env_full(enc, dec, store, device, batch=100, steps=20, train_iters=6):
    """configs[1] at its real size: student-forced rollouts over the 10 567-viewpoint table and all 90 graphs
    with the environment on the device (every next panorama depends on the action just chosen and is a
    cold 295 KB block of a 3.1 GB table): (a) argmax inference as a hipGraph replay, (b) the full TRAINING
    iteration with the reference's default `sample` feedback (train.py:299-300): rollout with dropout,
    BPTT, two Adam steps -- follower.py:1001-1020."""
    from . import nav, follower, dp, optim
    t0 = time.perf_counter()
    e, nt = full_world(store, batch)
    build_s = time.perf_counter() - t0
    e.reset_epoch()
    e._next_minibatch(True)
    items = list(e.batch)
    navb = nav.DeviceNavBatch(nt, items, steps)
    out = dict(what='student-forced rollouts on the FULL real environment: %d scans, %d viewpoints (feature table '
                    '%.2f GB), %d x 36 states x <=%d candidates tabulated on the device; batch %d from %d scans, '
                    '%d decode steps, encoder included; every step executed for every row'
                    % (len(nt.scans), nt.n_rows, store.table.numel() * 4 / 1e9, nt.n_rows, nt.A, batch,
                       len({it['scan'] for it in items}), steps),
               unit='agent-steps/s', host_build_seconds=build_s)
    enc.eval()
    dec.eval()
    eng = follower.FollowerEngine(enc, dec, store)
    replay, gst = eng.capture(navb, steps, 'argmax')
    dt = _timed(replay, 3, 10)
    stopped = float((gst.actions == 0).any(dim=0).float().mean())
    out['inference_argmax'] = dict(value=batch * steps / dt, ms_per_rollout=1e3 * dt,
                                   launch='hipGraph replay, one host sync per rollout',
                                   fraction_of_rows_that_stop=stopped)
    # (b) training iteration, sample feedback, on copies of the weights (the caller's models stay untouched)
    from . import model, synth
    d = synth.FULL
    enc2 = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc.embedding.weight.detach().cpu().numpy())
    dec2 = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc2.load_state_dict(enc.state_dict())
    dec2.load_state_dict(dec.state_dict())
    enc2.to(device).train()
    dec2.to(device).train()
    pe = [p for p in enc2.parameters() if p.requires_grad]
    pd = [p for p in dec2.parameters() if p.requires_grad]
    flat = dp.FlatGrads(pe + pd)
    oe = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4)
    od = optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
    eng2 = follower.FollowerEngine(enc2, dec2, store)

    def it():
        flat.zero()
        st = eng2.rollout(navb, steps, 'sample', train=True)
        st.loss.backward()
        oe.step()
        od.step()
        return st
    for _ in range(2):
        st = it()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(train_iters):
        st = it()
    torch.cuda.synchronize()
    dtt = (time.perf_counter() - t1) / train_iters
    # the same iteration as ONE hipGraph replay (runtime.TrainingGraph): the sampled actions, the environment steps they
    # drive and the dropout masks differ from replay to replay through device words only
    dtg = loss_g = None
    try:
        tg = eng2.capture_training(navb, steps, 'sample', optimizers=(oe, od), zero=flat)
        for _ in range(2):
            tg.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(train_iters):
            tg.replay()
        torch.cuda.synchronize()
        dtg = (time.perf_counter() - t1) / train_iters
        loss_g = float(tg.state.loss_buf)
    except Exception as exc:                     # (reported, not fatal: the eager figure stands)
        out['train_sample_feedback_graph_error'] = repr(exc)[:300]
    # (b') the same training through the agents' API (Seq2SeqAgent.train, follower.py:1001-1020) on a NEW minibatch every
    # iteration, host packing of the minibatch and the loss read included: replayed graphs (the default with
    # optim.FusedAdam on the device environment) and launch by launch
    from . import agents
    try:
        api = {}
        for mode in ('graph', 'eager'):
            e.reset_epoch()
            ag = agents.Seq2SeqAgent(e, '/tmp/sf_bench_agent_train.json', enc2, dec2, episode_len=steps)
            ag.store = store
            ag.use_device_env(nt)
            ag.train_graph = mode == 'graph'
            ag.train(oe, od, 3, feedback='sample')                  # warm-up (graph mode: one eager iteration + capture)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ag.train(oe, od, train_iters, feedback='sample')
            torch.cuda.synchronize()
            api[mode] = dict(ms_per_iteration=1e3 * (time.perf_counter() - t1) / train_iters, loss=float(ag.losses[-1]),
                             fallbacks=int(ag._engine.fallbacks))
        api['what'] = ('Seq2SeqAgent.train(FusedAdam, FusedAdam, n, feedback="sample") on the device environment, a new '
                       'minibatch of %d per iteration (packing, 7 small H2D copies and the loss read inside the time)' % batch)
        out['train_through_the_agent_api'] = api
    except Exception as exc:                     # (reported, not fatal)
        out['train_through_the_agent_api_error'] = repr(exc)[:300]
    # (b'') validation through the agents' API: Seq2SeqAgent.test (follower.py:987-999: argmax rollouts over a whole split,
    # the results dictionary) -- one inference graph replay per minibatch, the next minibatch encoded under it
    try:
        et, _ = full_world(store, batch, seed=41, n_items=20 * batch)
        ag = agents.Seq2SeqAgent(et, '/tmp/sf_bench_agent_test.json', enc, dec, episode_len=steps)
        ag.store = store
        ag.use_device_env(nt)
        ag.test(use_dropout=False, feedback='argmax')              # (first epoch: hop tables, the capture)
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            res = ag.test(use_dropout=False, feedback='argmax')
            torch.cuda.synchronize()
            dtt = time.perf_counter() - t1
            best = dtt if best is None else min(best, dtt)
        out['test_through_the_agent_api'] = dict(
            value=len(res) / best, unit='instructions/s', ms_per_minibatch=1e3 * best / (len(res) / batch),
            fallbacks=int(ag._engine.fallbacks),
            what='Seq2SeqAgent.test(feedback="argmax") over %d instructions in minibatches of %d on the device environment: '
                 'result dictionaries included (best of 3 epochs)' % (len(res), batch))
    except Exception as exc:                     # (reported, not fatal)
        out['test_through_the_agent_api_error'] = repr(exc)[:300]
    # (b''') configs[2] through the agents' API: Seq2SeqSpeaker.test (speaker.py:397-414, what
    # data_augmentation_from_speaker.py drives): greedy instructions for the gold paths of a split, routes from the
    # navigation tables, the split decoded as one two-stream sweep of replayed graphs
    try:
        senc, sdec = _speaker_models(store.device)
        es, _ = full_world(store, batch, seed=33, n_items=20 * batch)
        spk = agents.Seq2SeqSpeaker(es, '/tmp/sf_bench_speaker_test.json', senc, sdec, 80)
        spk.store = store
        spk.sweep_test_after = 0
        spk.test(use_dropout=False, feedback='argmax')             # (hop tables; the sweep's graphs)
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            res = spk.test(use_dropout=False, feedback='argmax')
            torch.cuda.synchronize()
            dtt = time.perf_counter() - t1
            best = dtt if best is None else min(best, dtt)
        out['speaker_test_through_the_agent_api'] = dict(
            value=len(res) / best, unit='paths/s', ms_per_minibatch=1e3 * best / (len(res) / batch),
            as_a_sweep='_test_sweep' in spk.__dict__,
            fallbacks=int(spk._test_sweep[1].fallbacks) if '_test_sweep' in spk.__dict__ else 0,
            what='Seq2SeqSpeaker.test(feedback="argmax") over %d gold paths in minibatches of %d, 80 words each: result '
                 'dictionaries included (best of 3 epochs)' % (len(res), batch))
    except Exception as exc:                     # (reported, not fatal)
        out['speaker_test_through_the_agent_api_error'] = repr(exc)[:300]
    # (c) configs[4] on the same world: state-factored search K = 40 over a minibatch of 64 instructions
    n_mb = 8
    e64, _ = full_world(store, 64, seed=15, n_items=64 * (n_mb + 2))
    agent = agents.Seq2SeqAgent(e64, '/tmp/sf_bench_search_full.json', enc, dec, episode_len=8)
    agent.store = store
    e64.set_beam_size(40)
    e64.reset_epoch()
    times, n_cand = [], 0
    for i in range(n_mb + 2):                    # two warm-up minibatches (graph capture, allocator), then the timed ones;
        torch.cuda.synchronize()                 # EVERY search runs on a minibatch the process has not seen before
        t2 = time.perf_counter()
        with torch.no_grad():
            trajs, _, _ = agent.state_factored_search(40, 1)
        torch.cuda.synchronize()
        if i >= 2:
            times.append(time.perf_counter() - t2)
            n_cand += sum(len(t_) for t_ in trajs)
    mean = sum(times) / len(times)
    out['state_factored_search_k40_b64'] = dict(
        value=64 / mean, unit='instructions/s', seconds=mean, seconds_best=min(times), seconds_worst=max(times),
        minibatches=n_mb, candidates_per_minibatch=n_cand / n_mb, episode_len=8,
        how='mean over %d DISTINCT minibatches of 64 instructions (states not seen before: nothing served from the sweep '
            'cache); per iteration one hipGraph replay (search.GraphStep) + one native bookkeeping call '
            '(sim/frontier_core.cpp)' % n_mb)
    out['train_sample_feedback'] = dict(value=batch * steps / dtt, ms_per_iteration=1e3 * dtt, iterations=train_iters,
                                        loss=float(st.loss.detach()),
                                        what='rollout with dropout 0.5 + sampled actions on the device env, BPTT, '
                                             '2x Adam; eager issue, same minibatch every iteration')
    if dtg is not None:
        out['train_sample_feedback']['graph_replay'] = dict(
            value=batch * steps / dtg, ms_per_iteration=1e3 * dtg, loss=loss_g,
            what='the same iteration as one hipGraph replay: fresh samples / masks per replay through device words')
    return out


@_guard
def speaker_train_iteration(store, device, batch=100, words=80, iters=10):
    """The speaker's training iteration (speaker.py:376-395 + train_speaker.py:28-31): teacher-forced scoring of a
    minibatch of `batch` paths x `words` words with dropout, backward, two Adam steps -- the word loop and its backward as
    one library call each (sf_speaker_words_fwd / _bwd), weight gradients as one product over all S*B rows."""
    from . import synth, speaker, optim, dp
    senc, sdec = _speaker_models(device)
    senc.train()
    sdec.train()
    n_vp = store.table.shape[0]
    sb = synth.speaker_batch(seed=0, batch=batch, n_viewpoints=n_vp, min_path=4, max_path=7, min_len=10, max_len=79)
    b = speaker.DeviceSpeakerBatch.from_synth(sb, device=device)
    pe = [p for p in senc.parameters() if p.requires_grad]
    pd = [p for p in sdec.parameters() if p.requires_grad]
    flat = dp.FlatGrads(pe + pd)
    oe, od = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4), optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
    eng = speaker.SpeakerEngine(senc, sdec, store)

    def it():
        flat.zero()
        st = eng.score(b, words, 'teacher', train=True)
        st.loss.backward()
        oe.step()
        od.step()
        return st
    for _ in range(10):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        st = it()
    host = (time.perf_counter() - t0) / iters
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    eager = dict(ms_per_iteration=1e3 * dt, ms_host_issue=1e3 * host, loss=float(st.loss.detach()))
    # the same iteration as ONE hipGraph replay (runtime.TrainingGraph: dropout sites and Adam steps are device words
    # written in front of each replay, so every replay is a new, valid iteration -- tests/test_gpu_training_graph.py)
    tg = eng.capture_training(b, words, optimizers=(oe, od))
    for _ in range(3):
        tg.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tg.replay()
    host_g = (time.perf_counter() - t0) / iters
    torch.cuda.synchronize()
    dt_g = (time.perf_counter() - t0) / iters
    return dict(what='speaker training iteration: %d paths x %d words, teacher forcing, dropout 0.5, BPTT, 2x Adam; ONE hipGraph '
                     'replay per iteration (fresh dropout sites / Adam steps through device words); `eager` = the same iteration '
                     'issued launch by launch' % (batch, words),
                value=batch * words / dt_g, unit='word-steps/s', ms_per_iteration=1e3 * dt_g, ms_host_issue=1e3 * host_g,
                loss=float(tg.state.loss_buf), eager=eager)


@_guard
def pragmatic_inference(enc, dec, store, device, instances=64, k=40, minibatches=6, profiler=None):
    """BASELINE configs[4] END TO END through the agents' API on the full world, per minibatch of `instances`
    instructions (rational_follower.py:35-148): Seq2SeqAgent.state_factored_search(K, 1), the speaker's teacher-forced
    score of EVERY candidate route (Seq2SeqSpeaker._score_obs_actions_and_instructions over all of them at once, as
    the reference does), rational_mix.  Mean over `minibatches` DISTINCT minibatches after two warm-up ones."""
    from . import agents, search
    e, _ = full_world(store, instances, seed=15, n_items=instances * (minibatches + 2))
    enc.eval()
    dec.eval()
    senc, sdec = _speaker_models(device)
    follower = agents.Seq2SeqAgent(e, '/tmp/sf_bench_pragmatic.json', enc, dec, episode_len=8)
    follower.store = store
    spk = agents.Seq2SeqSpeaker(e, '/tmp/sf_bench_pragmatic_speaker.json', senc, sdec, 80)
    spk.store = store
    follower.set_beam_size(k)
    follower.candidates_hook = spk.route_scores_hook('teacher')           # (what search.run_rational_follower sets up)
    e.reset_epoch()
    spk.score_marks = []
    follower.search_marks = []
    t_search, t_score, t_mix, n_cand = [], [], [], []
    phases, sphases = {}, {}
    for i in range(minibatches + 2):
        timed = i >= 2
        if timed and profiler is not None:
            profiler.enable()
        torch.cuda.synchronize()
        del follower.search_marks[:]
        t0 = time.perf_counter()
        with torch.no_grad():
            cands, hyps, walks = search._follower_candidates(follower, k, False, False, True, 4)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        flat = search.flatten(cands)
        del spk.score_marks[:]
        with torch.no_grad():
            spoken, _ = spk._score_obs_actions_and_instructions(
                [c['observations'] for c in flat], [c['actions'] for c in flat], [c['instr_encoding'] for c in flat],
                feedback='teacher')
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        by_id = {}
        for c, sp in zip(flat, spoken):
            c['follower_score'], c['speaker_score'] = c['score'], sp['score']
            by_id.setdefault(c['instr_id'], []).append(c)
        results, _ = search.rational_mix(by_id, 0.95)
        t3 = time.perf_counter()
        if timed:
            if profiler is not None:
                profiler.disable()
            t_search.append(t1 - t0)
            t_score.append(t2 - t1)
            t_mix.append(t3 - t2)
            n_cand.append(len(flat))
            for (_, a_), (name, b_) in zip(spk.score_marks, spk.score_marks[1:]):
                phases[name] = phases.get(name, 0.0) + (b_ - a_) / minibatches
            for (_, a_), (name, b_) in zip(follower.search_marks, follower.search_marks[1:]):
                sphases[name] = sphases.get(name, 0.0) + (b_ - a_) / minibatches
    total = [a + b + c for a, b, c in zip(t_search, t_score, t_mix)]
    ms = lambda x: 1e3 * sum(x) / len(x)                                   # noqa: E731
    return dict(what='pragmatic inference per minibatch of %d instructions on the full world (90 scans): '
                     'state_factored_search(K=%d, 1) + speaker teacher-forced scoring of every candidate route (one batch '
                     'of all of them) + rational_mix, through the agents\' API as search.run_rational_follower drives them (the '
                     'search hands its routes to the speaker in index form before it builds their result dictionaries: '
                     'ms_search includes issuing the scoring sweep, ms_speaker_scoring is what is left to collect); mean '
                     'over %d distinct minibatches' % (instances, k, minibatches),
                value=instances / (sum(total) / len(total)), unit='instructions/s', ms_per_minibatch=ms(total),
                ms_best=1e3 * min(total), ms_worst=1e3 * max(total), ms_search=ms(t_search), ms_speaker_scoring=ms(t_score),
                ms_rational_mix=ms(t_mix), candidates_per_minibatch=sum(n_cand) / len(n_cand),
                ms_search_phases={k_: round(1e3 * v, 2) for k_, v in sphases.items()},
                ms_speaker_scoring_phases={k_: round(1e3 * v, 2) for k_, v in phases.items()})


def _synthetic_states(rng, n, n_vp, a_max=14):
    """Index-form observations of `n` search states (env.R2RIndexEnv layout) with random candidates."""
    obs, udesc = [], []
    for _ in range(n):
        a_num = int(1 + np.clip(rng.poisson(4), 1, a_max - 1))
        adj = [dict(absViewIndex=-1, rel_heading=0.0, rel_elevation=0.0)]
        for _a in range(1, a_num):
            adj.append(dict(absViewIndex=int(rng.integers(0, 36)),
                            rel_heading=float(rng.uniform(-np.pi, np.pi)),
                            rel_elevation=float(rng.uniform(-np.pi / 6, np.pi / 6))))
        obs.append(dict(vp_row=int(rng.integers(0, n_vp)), viewIndex=int(rng.integers(0, 36)),
                        adj_loc_list=adj))
        udesc.append((int(rng.integers(0, n_vp)), int(rng.integers(0, 36)),
                      float(rng.uniform(-np.pi, np.pi)), float(rng.uniform(-np.pi / 6, np.pi / 6))))
    return obs, udesc


@_guard
def search_step(enc, dec, store, device, instances=64, k=40, words=80):
    from . import synth, search, speaker
    from .follower import batch_instructions_from_encoded
    rng = np.random.default_rng(7)
    n_vp = store.table.shape[0]
    instr = synth.instructions(3, instances, 10, 79, synth.FULL, sort=True)
    seq, mask, lengths = batch_instructions_from_encoded(instr, 80, reverse=True, device=device)
    with torch.no_grad():
        ctx, h_t, c_t = enc(seq, lengths)
    out = dict(what='one expansion step of the follower search over a flat state list: h/c row gather, '
                    'AttnDecoderLSTM step with per-state instruction rows, log-softmax + sorted top-k, '
                    'host packing of the index-form observations and D2H of the top-k included',
               unit='states/s')
    for label, n in (('state_factored_64', instances), ('beam_64x40', instances * k)):
        obs, udesc = _synthetic_states(rng, n, n_vp)
        rows = [int(i) for i in rng.integers(0, instances, size=n)]
        inst = [i % instances for i in range(n)]

        def step():
            fd = search.FlatDecoder(dec, store, ctx, mask)
            fd.seed(h_t, c_t)
            with torch.no_grad():
                fd.step(obs, udesc, rows, inst, k)
        dt = _timed(step, 2, 5)
        out[label] = dict(value=n / dt, ms_per_step=1e3 * dt, states=n,
                          how='dictionary observations packed on the host (the round-1 entry point FlatDecoder.step)')
        # what frontier.beam_search / state_factored_search issue since round 3: the same step over INDEX ARRAYS
        # (FlatDecoder.step_arrays: one upload, ~8 library calls, one download)
        a_max = 14
        a_num = (1 + np.clip(rng.poisson(4, n), 1, a_max - 1)).astype(np.int64)
        inp = dict(vp=rng.integers(0, n_vp, n), view=rng.integers(0, 36, n), a_num=a_num,
                   cand_view=rng.integers(0, 36, (n, a_max)), sincos=rng.standard_normal((n, a_max, 4)).astype(np.float32),
                   hrow=np.asarray(rows, np.int64), crow=np.asarray(inst, np.int64), has_u=rng.random(n) < 0.9,
                   u_vp=rng.integers(0, n_vp, n), u_view=rng.integers(0, 36, n),
                   u_sincos=rng.standard_normal((n, 4)).astype(np.float32))

        def step_a():
            fd = search.FlatDecoder(dec, store, ctx, mask)
            fd.seed(h_t, c_t)
            with torch.no_grad():
                fd.step_arrays(inp, k)
        dt_a = _timed(step_a, 2, 5)
        out[label]['index_arrays'] = dict(value=n / dt_a, ms_per_step=1e3 * dt_a,
                                          how='FlatDecoder.step_arrays: what the searches issue per step')
    # speaker rescoring of instances x k candidate paths (teacher-forced NLL of the instruction)
    senc, sdec = _speaker_models(device)
    n = instances * k
    chunk = 128
    # n / chunk DISTINCT candidate batches (ragged path lengths and instructions differ per chunk); the
    # timed region covers all of them: host packing + upload of each index batch and its teacher-forced
    # scoring (eager issue -- a graph per chunk would have to be re-captured for every new candidate set)
    sbs = [synth.speaker_batch(seed=100 + i, batch=chunk, n_viewpoints=n_vp, min_path=4, max_path=7, min_len=10,
                               max_len=79) for i in range(n // chunk)]
    eng = speaker.SpeakerEngine(senc, sdec, store)

    def rescore():
        with torch.no_grad():
            scores = [eng.score(speaker.DeviceSpeakerBatch.from_synth(sb, device=device), words, 'teacher').step_scores.sum(0)
                      for sb in sbs]
        return torch.cat(scores)
    dt = _timed(rescore, 1, 3)
    # for reference: ONE captured batch replayed (no packing, no upload), scaled to the same candidate count
    replay, _ = eng.capture(speaker.DeviceSpeakerBatch.from_synth(sbs[0], device=device), words, 'teacher')
    dt_graph = _timed(replay, 2, 5) * (n / chunk)
    out['speaker_rescoring_64x40'] = dict(
        value=n / dt, unit='candidates/s', ms_total=1e3 * dt,
        how='MEASURED: %d distinct teacher-forced batches of %d paths x %d words, host packing + upload + eager '
            'scoring of every batch inside the timed region' % (n // chunk, chunk, words),
        ms_total_one_graph_extrapolated=1e3 * dt_graph,
        extrapolated_how='one captured batch replayed, x %d (an extrapolation, not a measurement)' % (n // chunk))
    return out
