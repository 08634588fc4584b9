#!/usr/bin/env python3
"""Headline benchmark: follower rollout throughput in agent-steps/s (BASELINE.json).

One "step" of this script = one follower episode batch on every rank: EncoderLSTM over the
(<=80-token) instructions + `--decode-steps` AttnDecoderLSTM steps with on-device argmax
feedback, masking, cross-entropy and u_prev gather, over index-form observations gathered
from the HBM-resident 36x2048 feature table.  With --workload train the step also runs
BPTT, the (data-parallel) gradient all-reduce and two Adam updates (train.py:263-268).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 ...          # spawns 8 ranks itself (one process per GPU, RCCL)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # or be launched

Prints ONE JSON line on rank 0.  value = B * decode_steps * world * K / max-over-ranks time.

Process model: when WORLD_SIZE is not in the environment and --gpus N > 1, this process is only a
launcher -- it starts N children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on
127.0.0.1) BEFORE anything touches the GPU, waits for them and exits with their status.  It never
re-execs itself.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (kernel arguments in device memory: speaker_follower_amd/__init__.py; set here too because this script imports torch first)
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

# algorithmic work of ONE agent-step (forward, fp32, A = 8, L = 80, B = 100)
AGENT_STEP_FLOPS = 71386112.0
AGENT_STEP_BYTES = 1049469.0
# what the kernels EXECUTE per agent-step after the folds of DESIGN.md section 3 (the [36 x 2176] x [2176 x 256] key
# projection and the per-candidate action projection are re-associated away): gate product 19.92 M + t_v 0.26 +
# q 1.11 + visual attention 0.31 + t_text 0.52 + text attention 0.16 + h~ 1.05 + t_a 0.26 + r 1.11 + scores 0.03
def executed_agent_step_flops(H=512, F=2176, D=256, V=36, L=80, A=8):
    return 2.0 * ((2 * F + H) * 4 * H + H * D + D * F + 2 * V * F + H * H + 2 * L * H + 2 * H * H + H * D + D * F + A * F)


PEAK_TFLOPS_F32_MFMA = 157.3        # MI355X_MICROARCH.md: dense fp32 MFMA at 2.4 GHz
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s measured achievable)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', choices=['rollout', 'train'], default='rollout')
    ap.add_argument('--batch', type=int, default=100)
    ap.add_argument('--decode-steps', type=int, default=20)
    ap.add_argument('--n-viewpoints', type=int, default=10567)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip every measurement that is not `value`, `roofline` or `cpu_baseline`')
    ap.add_argument('--no-train-extra', action='store_true', help='skip the extra training-iteration measurement')
    ap.add_argument('--no-graph', action='store_true', help='issue the rollout eagerly instead of replaying a hipGraph')
    ap.add_argument('--row-shards', type=int, default=1,
                    help='rollout only: run the batch as this many concurrent row shards (measured: no gain, '
                         'a half-batch chain is as long as a full one)')
    ap.add_argument('--in-flight', type=int, default=2,
                    help='extra measurement: this many independent batch-100 rollouts in flight on separate streams')
    ap.add_argument('--cpu-reps', type=int, default=5, help='timed CPU-port rollouts behind one warm-up (median reported)')
    ap.add_argument('--two-stream-forward', action='store_true', help='(experiment) visual half on a side stream, device-flag ordering')
    # test-only switches: exercise the N > 1 code path (launcher, rendezvous, collectives, JSON) on a
    # box with ONE GPU.  Ranks share cuda:0 and reduce through gloo; the line is marked oversubscribed.
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl')
    ap.add_argument('--share-gpu', action='store_true', help='(test) every rank uses cuda:0')
    ap.add_argument('--force-collectives', action='store_true',
                    help='(test) --gpus 1 only: initialise the process group (world size 1) and issue every collective of '
                         'the data-parallel path anyway -- RCCL executes on this GPU; adds `train_dp` to the line')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help='weak (default): every rank takes its own batch of --batch rows; strong: ONE global batch of --batch '
                         'rows (the reference\'s batch is 100 globally, train.py:28) split over the ranks with dp.shard_rows')
    ap.add_argument('--extras-budget', type=int, default=240,
                    help='N > 1: seconds the data-parallel training extra may take before the headline line is printed without it')
    ap.add_argument('--extras-out', default=os.path.join(ROOT, 'bench_extras.json'),
                    help='file that receives the FULL result object (per-kernel tables, the other BASELINE configs, the '
                         'training iteration); stdout carries only the compact headline line (< 4 KB)')
    ap.add_argument('--plan', action='store_true',
                    help='print the child commands / environment `--gpus N` would start (JSON) and exit; touches no GPU')
    return ap.parse_args(argv)


def _shard(n_rows, rank, world):
    """dp.shard_rows without importing the package (the --plan path touches neither torch nor the GPU)."""
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


# ------------------------------------------------------------------------------------------ launcher
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_plan(n, argv, env=None, port=None):
    """The N child processes of `bench.py --gpus N`: [(command, environment)], one per GPU."""
    env = dict(os.environ if env is None else env)
    port = port or free_port()
    plan = []
    for r in range(n):
        e = dict(env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL needs it on this stack)
        plan.append(([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), e))
    return plan


def run_launcher(args, argv):
    """Parent of an N-GPU run.  Touches no GPU; children inherit stdout/stderr (only rank 0 prints)."""
    procs = [subprocess.Popen(cmd, env=env) for cmd, env in launch_plan(args.gpus, argv)]
    rc = 0
    try:
        pending = set(range(len(procs)))
        while pending:
            for i in sorted(pending):
                code = procs[i].poll()
                if code is None:
                    continue
                pending.discard(i)
                if code != 0 and rc == 0:
                    rc = code
                    for j in pending:                 # one rank failed: stop the others (exact PIDs)
                        procs[j].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------- workload
def build_models(seed, device):
    import torch
    from speaker_follower_amd import synth, model
    d = synth.FULL
    # 'peaky' gains: logit std ~1.4, attention maxima 0.6-0.8 -- with the flat initialisation every logit is ~0,
    # the loss is sum log(a_num) and a parity check against the CPU port says nothing (round-2 verdict)
    enc_w, dec_w = synth.follower_weights_peaky(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return enc.to(device), dec.to(device), enc_w, dec_w


def device_table(n_vp, seed, device):
    """ResNet-pool5-like table generated on the device (0.5*N(0,1) clipped at 0), 3.1 GB at
    the full 10 567 viewpoints."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.empty(n_vp, 36, 2048, device=device, dtype=torch.float32)
    chunk = 512
    for i in range(0, n_vp, chunk):
        blk = t[i:i + chunk]
        blk.normal_(0.0, 0.5, generator=g)
        blk.clamp_(min=0.0)
    return t


def cpu_baseline(enc_w, dec_w, fb, table_rows, row_of, decode_steps, reps, threads):
    """The numpy oracle (a port of the reference modules, oracle/np_model.py) on `threads` host
    threads (None = every core the BLAS pool takes), on the same batch the GPU ran."""
    import copy
    import numpy as np
    from threadpoolctl import threadpool_limits
    from oracle import np_env, np_model
    fbc = copy.copy(fb)
    fbc.vp = np.vectorize(row_of.get)(fb.vp).astype(np.int32)
    loc = np_env.static_loc_embeddings()
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, 80, reverse=True)
    B = len(lens)
    rates, res = [], None
    with threadpool_limits(limits=threads):
        for rep in range(reps + 1):           # one warm-up, then `reps` timed, MEDIAN
            t0 = time.perf_counter()
            res = np_model.follower_rollout(
                enc_w, dec_w, seq, lens, mask, decode_steps,
                lambda t: np_env.dense_follower_step(table_rows, loc, fbc, t), fb.target, 'argmax',
                2176, early_exit=False)       # same work as the GPU: every step for every row
            dt = time.perf_counter() - t0
            n = len(res['logits'])
            if rep > 0:
                rates.append(B * n / dt)
    cores = threads if threads else os.cpu_count()
    return dict(value=float(np.median(rates)), unit='agent-steps/s', cores=cores, kind='port',
                sample='%d full rollouts of the same batch (B=%d, %d decode steps, encoder included), '
                       'numpy oracle, %d thread%s: 1 warm-up + %d timed, median (min %.0f, max %.0f)'
                       % (reps + 1, B, n, cores, '' if cores == 1 else 's (BLAS pool)', reps, min(rates), max(rates))), res


def measure_train(enc, dec, store, batch, S, iters, warmup, group=None, world=1, coll=None, global_rows=None):
    """follower.py:1001-1020 + train.py:263-268 per iteration: zero_grad, student-forcing rollout with
    loss, backward, [gradient all-reduce,] Adam(lr 1e-4, weight_decay 5e-4) on encoder and decoder.
    `global_rows`: rows of the whole job per iteration (strong scaling: `batch` is this rank's shard of ONE global
    batch); default = batch rows x world (weak scaling)."""
    import torch
    from speaker_follower_amd import follower, dp, optim
    coll = world > 1 if coll is None else coll       # collectives issued (world > 1, or forced at one rank)
    enc.train()
    dec.train()
    params_e = [p for p in enc.parameters() if p.requires_grad]
    params_d = [p for p in dec.parameters() if p.requires_grad]
    # gradients: ONE buffer laid out in the order the backward completes them (decoder LSTM 40 MB first, the
    # other decoder weights, the encoder last); with several ranks each bucket's all-reduce is launched from
    # the backward as soon as the launches that complete it are issued (dp.BucketedGrads)
    flat = dp.BucketedGrads(dp.follower_buckets(enc, dec), group=group)
    opt_e = optim.FusedAdam(params_e, lr=1e-4, weight_decay=5e-4)      # one launch per step each
    opt_d = optim.FusedAdam(params_d, lr=1e-4, weight_decay=5e-4)
    engine = follower.FollowerEngine(enc, dec, store, group=group)
    B = batch.batch_size
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    mode = dict(sync='buckets')          # 'buckets' (overlapped), 'blocking' (one all-reduce after backward), 'none'

    def it(k=None):
        flat.zero()
        engine.grad_sync = flat if (coll and mode['sync'] == 'buckets') else None
        st = engine.rollout(batch, S, 'argmax', train=True)
        st.loss.backward()
        if mode['sync'] == 'buckets':
            if coll:
                flat.wait()
        elif mode['sync'] == 'blocking':
            if k is not None:
                ev[k][0].record()
            flat.allreduce(group)
            if k is not None:
                ev[k][1].record()
        opt_e.step()
        opt_d.step()
        return st

    def barrier():
        if coll:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    for _ in range(warmup):
        it()
    barrier()
    t0 = time.perf_counter()
    for k in range(iters):
        st = it(k)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if coll:
        tt = torch.tensor([dt], device=store.device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    dt /= iters
    ar = dict(allreduce_ms=0.0)
    if coll:
        # the same iteration (a) with ONE blocking all-reduce of the whole buffer behind backward() -- its
        # duration is the TOTAL all-reduce time -- and (b) with no gradient exchange at all: what the overlapped
        # buckets leave EXPOSED is (overlapped iteration) - (b)
        def timed(sync, n):
            mode['sync'] = sync
            it()
            barrier()
            t1 = time.perf_counter()
            for k in range(n):
                it(k)
            barrier()
            tt = torch.tensor([(time.perf_counter() - t1) / n], device=store.device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            return float(tt)
        n2 = min(iters, 4)
        dt_block = timed('blocking', n2)
        total_ms = sum(a.elapsed_time(b) for a, b in ev[:n2]) / n2
        dt_none = timed('none', n2)
        mode['sync'] = 'buckets'
        ar = dict(allreduce_ms=total_ms, allreduce_total_ms_blocking=total_ms,
                  allreduce_exposed_ms_overlapped=max(0.0, 1e3 * (dt - dt_none)),
                  ms_per_iteration_blocking_allreduce=1e3 * dt_block, ms_per_iteration_no_exchange=1e3 * dt_none,
                  buckets_bytes=[4 * (hi - lo) for lo, hi in flat.bounds],
                  schedule='buckets in production order (decoder LSTM, other decoder weights, encoder), each all-reduce '
                           'launched async behind the launches that complete it; wait before Adam')
    if coll:
        # the SAME data-parallel iteration as replayed graph segments (runtime.TrainingGraph, segmented: the capture is cut
        # at every collective point; a replay = segment, all-reduce start, segment, ...): what a data-parallel job runs
        try:
            engine.grad_sync = flat
            enc.train()
            dec.train()
            tg, why = None, None
            try:
                tg = engine.capture_training(batch, S, 'argmax', optimizers=(opt_e, opt_d), zero=flat)
            except Exception as exc:                        # noqa: BLE001
                why = repr(exc)[:300]
            # every rank replays or none does (a replay issues collectives: one rank that failed to capture would leave
            # the others waiting in an all-reduce)
            okf = torch.tensor([1.0 if tg is not None else 0.0], device=store.device)
            torch.distributed.all_reduce(okf, op=torch.distributed.ReduceOp.MIN)
            if float(okf) < 1.0:
                raise RuntimeError(why or 'another rank could not capture the segmented iteration')
            for _ in range(2):
                tg.replay()
            barrier()
            t1 = time.perf_counter()
            for _ in range(iters):
                tg.replay()
            barrier()
            tt = torch.tensor([(time.perf_counter() - t1) / iters], device=store.device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            ar['ms_per_iteration_graph_segments'] = 1e3 * float(tt)
            ar['graph_segments'] = len(tg.segments)
            ar['loss_graph_segments'] = float(tg.state.loss_buf)
        except Exception as exc:                            # (reported; the eager figures above stand)
            ar['graph_segments_error'] = repr(exc)[:300]
        engine.grad_sync = None
    # the fault words of the persistent launches (0 = healthy), and the production entry point once: FollowerEngine.run
    # (rollout + backward + fault check, the flag reduced over the group, per-step re-issue on a fault) -- per rank
    from speaker_follower_amd import runtime as _rt
    faults = _rt.take_fault(store.device)
    health = None
    if coll:
        engine.grad_sync = flat
        for _ in range(2):
            flat.zero()
            engine.run(batch, S, 'argmax', train=True, backward=True)
            flat.wait()
        engine.grad_sync = None
        torch.cuda.synchronize()
        mine = torch.tensor([faults, engine.fallbacks], device=store.device, dtype=torch.int32)
        every = [torch.zeros_like(mine) for _ in range(torch.distributed.get_world_size(group))]
        torch.distributed.all_gather(every, mine, group=group)
        health = dict(persistent_launch_faults=[int(t[0]) for t in every], fallbacks=[int(t[1]) for t in every],
                      note='per rank; faults = fault words raised by the timed iterations (0 = every persistent launch '
                           'completed), fallbacks = iterations FollowerEngine.run re-issued on the per-step kernels')
    kernels = None
    if world == 1 and not coll:   # per-kernel table of the iteration, measured in this run (both backward streams)
        from speaker_follower_amd import bench_extras
        rows, us = bench_extras.kernel_table(lambda: it())
        H, F, D = 512, 2176, 256
        executed = 3.0 * executed_agent_step_flops(H, F, D, 36, max(batch.lengths), 5.0) * B * S
        kernels = dict(rows=rows, kernel_time_ms_per_iteration=1e-3 * us,
                       executed_gflop=executed / 1e9, executed_flops_frac=executed / dt / 1e12 / PEAK_TFLOPS_F32_MFMA,
                       note='kernel times add up to more than the iteration: the backward runs on two streams; '
                            'executed FLOPs = 3 x the forward (data + weight gradients), encoder excluded')
    enc.eval()
    dec.eval()
    graph = None
    if world == 1 and not coll:
        # the same iteration as ONE hipGraph replay (runtime.TrainingGraph): the dependent chain without launch gaps
        engine.grad_sync = None
        enc.train()
        dec.train()
        tg = engine.capture_training(batch, S, 'argmax', optimizers=(opt_e, opt_d), zero=flat)
        for _ in range(3):
            tg.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(iters):
            tg.replay()
        torch.cuda.synchronize()
        dt_g = (time.perf_counter() - t1) / iters
        graph = dict(ms_per_iteration=1e3 * dt_g, value=B * S / dt_g, unit='agent-steps/s', loss=float(tg.state.loss_buf),
                     what='the same iteration as one hipGraph replay (dropout sites / Adam steps through device words: every '
                          'replay is a new valid iteration, bit-identical to the eager loop)')
        enc.eval()
        dec.eval()
    rows = B * world if global_rows is None else global_rows
    if graph is not None:
        # one process, no collectives: the iteration as it would be run in production is ONE graph replay; the eager
        # issue of the same launches is reported beside it
        eager = dict(ms_per_iteration=1e3 * dt, value=rows * S / dt, what='the same iteration issued launch by launch')
        dt_pub = 1e-3 * graph['ms_per_iteration']
        return dict(value=rows * S / dt_pub, unit='agent-steps/s', ms_per_iteration=1e3 * dt_pub, iterations=iters,
                    scaling='weak', global_batch=rows, rows_this_rank=B, roofline=kernels, health=health, eager=eager,
                    launch='hipGraph replay (runtime.TrainingGraph): dropout sites and Adam steps are device words written '
                           'in front of each replay; bit-identical to the eager loop (tests/test_gpu_training_graph.py)',
                    allreduce_bytes=flat.flat.numel() * 4, **ar,
                    what='student-forcing rollout (dropout 0.5) + BPTT + 2x Adam (one HIP launch each), batch %d per GPU, '
                         '%d decode steps' % (B, S), loss=graph['loss'])
    return dict(value=rows * S / dt, graph=graph, unit='agent-steps/s', ms_per_iteration=1e3 * dt, iterations=iters,
                scaling='weak' if global_rows is None else 'strong', global_batch=rows, rows_this_rank=B,
                roofline=kernels, health=health,
                allreduce_bytes=flat.flat.numel() * 4, **ar,
                what='student-forcing rollout (dropout 0.5) + BPTT + %s2x Adam (one HIP launch each), %s, '
                     '%d decode steps, eager issue' % ('bucketed sum all-reduce over %d ranks overlapped with the backward + ' % world
                                                       if world > 1 else '',
                                                       'batch %d per GPU' % B if global_rows is None else
                                                       'ONE global batch of %d rows split over %d ranks (dp.shard_rows; %d here)'
                                                       % (rows, world, B), S),
                loss=float(st.loss.detach()))


# --------------------------------------------------------------------------- per-kernel roofline table
def kernel_work(name, B, S, T, A_mean, dims):
    """Algorithmic (flops, bytes, what) of ONE launch of a forward-rollout kernel, from the shapes
    alone (fp32).  Bytes = every operand and result once, weights included (SURVEY 8d convention)."""
    H, F, D, V = dims
    f4 = 4.0
    small = lambda M, N, K: (2.0 * M * N * K, f4 * (N * K + M * K + M * N))     # noqa: E731  y = x W^T
    if 'enc_persist_kernel' in name:
        return (T * 2.0 * B * H * 4 * H, f4 * (4 * H * H + T * B * (H + 4 * H + 4 * H + 3 * H)),
                'all %d encoder recurrent steps in one persistent launch (W_hh register-resident, batch rows '
                'partitioned across the XCDs)' % T)
    if 'lstm_step' in name:
        return (2.0 * B * H * 4 * H, f4 * (4 * H * H + B * (H + 4 * H + 4 * H)),
                'encoder recurrent step: gates = h W_hh^T + table row, cell update')
    if 'gemm_nt_split' in name:
        return (2.0 * B * (2 * F + H) * 4 * H, f4 * (4 * H * (2 * F + H) + B * (2 * F + H) + B * 4 * H),
                'decoder LSTMCell gate product [B,2F+H] x [4H,2F+H]^T, fp32 operands split error-free into 3 bf16 pieces, '
                '6 v_mfma_f32_16x16x32_bf16 per product with fp32 accumulate (fp32-class accuracy, measured closer to '
                'float64 than the fp32 MFMA); priced as its ALGORITHMIC fp32 FLOPs against the fp32 MFMA peak -- it '
                'executes 6x that many bf16 FLOPs = %.2f of the 2.5 PFLOP/s bf16 peak at this launch time')
    if 'gemm_nt_tiled' in name:
        return (2.0 * B * (2 * F + H) * 4 * H, f4 * (4 * H * (2 * F + H) + B * (2 * F + H) + B * 4 * H),
                'decoder LSTMCell gate product [B,2F+H] x [4H,2F+H]^T')
    # ---- the folded inference chain (round 6: text attention over ctx W_in / ctx W_out[:, :H]^T)
    if 'pair_textfold_small_small' in name:
        fl, by = small(B, H, H)
        fl2, by2 = small(B, D, H)
        return (fl + fl2 + 4.0 * B * T * H, by + by2 + f4 * 2 * B * T * H,
                'folded text attention (scores over ctx W_in, sum over ctx W_out1^T; 4 groups per sample merged in the '
                'launch) || y = W_out[:, H:] h1 || t_v = W_h h1 + b')
    if 'pair_apro_small' in name:
        fl, by = small(B, D, H)
        fl2, by2 = small(B, F, D)
        return (fl + fl2, by + by2 + f4 * B * H, 't_a = W_h tanh(z + y) + b (A-prologue) || q = W_v^T t_v')
    if 'pair_vis_small_kernel<4, 2' in name or 'pair_vis_small_kernel<2, 2' in name or 'pair_vis_small_kernel<1, 2' in name:
        fl, by = small(B, F, D)
        return (fl + 4.0 * B * V * F, by + f4 * B * V * F,
                'visual-attention partials of step t+1 (panorama rows read once) || r = W_a^T wt')
    if 'pair_score_merge' in name:
        return (2.0 * B * A_mean * F + 2.0 * B * F, f4 * B * (A_mean + 1) * F + f4 * B * F + f4 * 3 * B * F,
                'candidate rows . r, mask, CE, argmax, u_next gather || merge of the attention partials')
    if 'gemm_nt_big_kernel' in name:
        return (2.0 * B * T * H * H, f4 * (H * H + 2 * B * T * H),
                'once per episode: ctx W_in and ctx W_out[:, :H]^T ([B T, H] x [H, H], many-row bf16x6 kernel)')
    if 'pair_vis_small_kernel<1, 8' in name:
        fl, by = small(B, H, 2 * H)
        return (fl + 4.0 * B * V * F, by + f4 * B * V * F,
                'visual-attention partials of step t+1 (panorama rows read once) || h~ = tanh(W_out [wc;h])')
    if 'pair_vis_small_kernel<1, 4' in name:
        fl, by = small(B, D, H)
        return (fl + 2.0 * B * F, by + f4 * 2 * B * F, 'merge of the attention partials || t_a = W_h h~ + b')
    if 'pair_small_text' in name:
        fl, by = small(B, F, D)
        return (fl + 4.0 * B * T * H, by + f4 * B * T * H, 'text attention over ctx [B,L,H] || q = W_v^T t_v')
    if 'pair_small_small' in name:
        fl, by = small(B, H, H)
        fl2, by2 = small(B, D, H)
        return (fl + fl2, by + by2, 't_text = W_in h1 || t_v = W_h h1 + b')
    if 'score_glue' in name:
        return (2.0 * B * A_mean * F, f4 * B * (A_mean + 1) * F + f4 * B * F,
                'candidate rows . r, mask, CE, argmax, u_next gather')
    if 'lstm_pw_fwd' in name:
        return (10.0 * B * 4 * H, f4 * B * (8 * 4 * H + 4 * H + 4 * H), 'split-K slab sum + LSTM cell update')
    if 'gemm_nt_small_kernel<4, 2>' in name:
        fl, by = small(B, F, D)
        return (fl, by, 'r = W_a^T wt')
    return (None, None, '')


def mfma_counters(workload='rollout'):
    """{kernel name: matrix-pipe busy cycles per SIMD per dispatch} from the committed rocprofv3 --pmc pass
    (profiles/pmc_mfma.json <- profiles/r06_a_mfma_counters.txt; tools/pmc_mfma.sh).  OFFLINE constants of the batch-100
    default workload; divided by a launch time measured in THIS run they give `mfma_busy_frac`."""
    path = os.path.join(ROOT, 'profiles', 'pmc_mfma.json')
    if not os.path.exists(path):
        return {}, 2.4
    d = json.load(open(path))
    return ({k: v[workload]['mfma_busy_cycles_per_simd'] for k, v in d['kernels'].items() if workload in v},
            float(d.get('clock_ghz', 2.4)))


def roofline_table(prof_rows, n_rollouts, B, S, T, A_mean, dims, pmc, busy=None, clock_ghz=2.4):
    """Top kernels of the profiled eager rollouts, each priced against both ceilings."""
    busy = busy or {}
    total = sum(r['total_us'] for r in prof_rows.values())
    out = []
    for name, r in sorted(prof_rows.items(), key=lambda kv: -kv[1]['total_us']):
        fl, by, what = kernel_work(name, B, S, T, A_mean, dims)
        row = dict(kernel=name, calls_per_rollout=r['calls'] / n_rollouts, avg_us=r['avg_us'],
                   share=r['total_us'] / total, what=what)
        if fl is not None:
            if '%.2f' in what:
                what = what % (6.0 * fl / (r['avg_us'] * 1e-6) / 2.5e15)
                row['what'] = what
                # the same launch priced as what it EXECUTES: 6 bf16 MFMA products per fp32 product, dense bf16 peak
                row['executed'] = dict(dtype='bf16 (3-way error-free split of fp32 operands, fp32 accumulate)',
                                       tflops=6.0 * fl / (r['avg_us'] * 1e-6) / 1e12, peak=2500.0,
                                       frac=6.0 * fl / (r['avg_us'] * 1e-6) / 2.5e15)
            tf = fl / (r['avg_us'] * 1e-6) / 1e12
            gbs = by / (r['avg_us'] * 1e-6) / 1e9
            row.update(flops_per_launch=fl, bytes_per_launch=by, tflops=tf, mfma_frac=tf / PEAK_TFLOPS_F32_MFMA,
                       hbm_gbs=gbs, hbm_frac=gbs / PEAK_HBM_GBS)
        for key, v in pmc.items():
            if key in name:
                row['traffic_offline_pmc'] = v
        for key, cyc in busy.items():
            if name == key:
                # SQ_VALU_MFMA_BUSY_CYCLES per SIMD (counter, offline) / this run's launch time in shader cycles
                row['mfma_busy_frac'] = cyc / (r['avg_us'] * 1e-6 * clock_ghz * 1e9)
                row['mfma_busy_cycles_per_simd'] = cyc
        out.append(row)
    return out, total



# ------------------------------------------------------------------------------------------ the printed line
HEADLINE_LIMIT = 4096        # bytes; the driver reads ONE line from stdout, and a 20 KB line was once lost on the way


def _cut(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + '...'


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def headline_of(full, extras_file):
    """The compact object that is printed: the contract keys, the dominant kernel's roofline (both prices: the
    algorithmic fp32 work against the fp32 MFMA peak AND what the launch executes on the bf16 matrix cores against the
    bf16 peak), the CPU baseline, the parity check of this run, and a few scalars of the extras.  Everything else
    (per-kernel tables, prose, the other configs) is in `extras_file`."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data')
    h = {k: full[k] for k in keep if k in full}
    cfg = dict(full.get('config', {}))
    cfg['workload'] = _cut(cfg.get('workload', ''), 260)
    h['config'] = cfg
    r = full.get('roofline')
    if r:
        rr = _pick(r, ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'launch_us', 'flops_per_launch',
                       'bytes_per_launch', 'mfma_frac', 'hbm_frac', 'mfma_busy_frac', 'kernel_time_ms_per_rollout'))
        rr.setdefault('traffic', None)
        rr['kernel'] = _cut(r.get('kernel', '').split(' (')[0], 80)
        if r.get('executed'):
            rr['executed'] = _pick(r['executed'], ('dtype', 'tflops', 'peak', 'frac'))
            rr['executed']['dtype'] = 'bf16'
        if r.get('rollout'):
            rr['rollout'] = _pick(r['rollout'], ('flops_frac', 'executed_flops_frac', 'hbm_frac'))
        rr['kernels'] = [_pick(k, ('kernel', 'avg_us', 'share', 'mfma_frac', 'hbm_frac', 'mfma_busy_frac'))
                         for k in r.get('kernels', [])[:5]]
        for k in rr['kernels']:
            k['kernel'] = _cut(k['kernel'], 48)
        rr['note'] = ('frac = algorithmic fp32 FLOPs / launch time / fp32 MFMA peak; executed = the 6 bf16 MFMA products '
                      'per fp32 product the launch runs, against the dense bf16 peak; mfma_busy_frac = '
                      'SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES from the committed rocprofv3 --pmc pass')
        h['roofline'] = rr
    if full.get('cpu_baseline'):
        c = _pick(full['cpu_baseline'], ('value', 'unit', 'cores', 'kind', 'sample'))
        c['sample'] = _cut(c.get('sample', ''), 200)
        h['cpu_baseline'] = c
        h['speedup_vs_cpu_1_core'] = full['value'] / c['value'] if c.get('value') else None
    if full.get('parity_vs_cpu_port'):
        h['parity_vs_cpu_port'] = _pick(full['parity_vs_cpu_port'],
                                        ('actions_bit_exact', 'max_abs_logit_diff', 'loss_abs_diff', 'max_abs_logit'))
    g9 = full.get('speaker_parity_g9')
    if isinstance(g9, dict) and 'argmax' in g9:
        h['parity_speaker_g9'] = {fb: _pick(g9[fb], ('words_bit_exact', 'vs_fp32_reference', 'vs_float64',
                                                       'reference_fp32_vs_its_float64')) for fb in ('argmax', 'teacher')}
    for k in ('launch', 'persistent_launch_faults', 'oversubscribed'):
        if k in full:
            h[k] = _cut(full[k], 120) if isinstance(full[k], str) else full[k]
    ex = {}
    for name, keys in (('concurrent', ('value', 'rollouts_in_flight')),
                       ('train_iteration', ('value', 'ms_per_iteration')),
                       ('speaker_decode', ('value', 'unit', 'ms_per_batch')),
                       ('speaker_sweep', ('value', 'unit', 'seconds')),
                       ('speaker_train_iteration', ('value', 'unit', 'ms_per_iteration')),
                       ('search_step', ('value', 'unit')),
                       ('real_env_full', ('value', 'unit')),
                       ('pragmatic_inference', ('value', 'unit')),
                       ('cpu_baseline_all_cores', ('value', 'cores'))):
        v = full.get(name)
        if isinstance(v, dict):
            if 'error' in v:
                ex[name] = dict(error=_cut(v['error'], 80))
            elif 'value' in v:
                ex[name] = _pick(v, keys)
    if ex:
        h['extras'] = ex
    td = full.get('train_dp')
    if isinstance(td, dict):
        keys = ('value', 'unit', 'ms_per_iteration', 'ms_per_iteration_graph_segments', 'allreduce_ms',
                'allreduce_exposed_ms_overlapped', 'ms_per_iteration_no_exchange', 'allreduce_bytes', 'global_batch',
                'scaling', 'graph_segments_error')
        c = _pick(td, keys)
        if 'error' in td:
            c['error'] = _cut(td['error'], 200)
        if isinstance(td.get('strong'), dict):
            c['strong'] = _pick(td['strong'], keys)
        if isinstance(td.get('health'), dict):
            c['faults'] = sum(td['health'].get('persistent_launch_faults', []))
        h['train_dp'] = c
    h['extras_file'] = extras_file
    return h


def print_result(full, result_out, extras_path):
    """Write the full object to `extras_path` (best effort) and print the compact headline as ONE line on the
    original stdout.  NaN / Infinity would make the line invalid JSON: refuse them (allow_nan=False), falling back to a
    line that says which part was not finite."""
    written = None
    for path in (extras_path, os.path.join(ROOT, 'gpurun_out', 'bench_extras.json')):
        try:
            if path and os.path.isdir(os.path.dirname(path)):
                with open(path, 'w') as f:
                    json.dump(full, f, indent=1, default=str)
                written = written or os.path.relpath(path, ROOT)
        except OSError:
            pass
    def shorten(o):                     # six significant digits are more than any of these measurements carries
        if isinstance(o, float) and o == o and o not in (float('inf'), float('-inf')):
            return float('%.6g' % o)
        if isinstance(o, dict):
            return {k: shorten(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [shorten(v) for v in o]
        return o
    h = shorten(headline_of(full, written))
    r = h.get('roofline')
    if isinstance(r, dict) and r.get('peak'):
        r['frac'] = r['achieved'] / r['peak']            # (exactly the quotient of the two numbers PRINTED beside it)
        if isinstance(r.get('executed'), dict) and r['executed'].get('peak'):
            r['executed']['frac'] = r['executed']['tflops'] / r['executed']['peak']
    try:
        line = json.dumps(h, allow_nan=False)
    except ValueError:
        def clean(o):
            if isinstance(o, float) and (o != o or o in (float('inf'), float('-inf'))):
                return None
            if isinstance(o, dict):
                return {k: clean(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [clean(v) for v in o]
            return o
        h = clean(h)
        h['non_finite_values_replaced_by_null'] = True
        line = json.dumps(h, allow_nan=False)
    if len(line) >= HEADLINE_LIMIT:          # never again a line the driver cannot hold: drop the optional parts
        for k in ('extras', 'train_dp'):
            h.pop(k, None)
        if 'roofline' in h:
            h['roofline'].pop('kernels', None)
            h['roofline'].pop('note', None)
        line = json.dumps(h, allow_nan=False)
    assert len(line) < HEADLINE_LIMIT, len(line)
    print(line, file=result_out)
    result_out.flush()
    return h


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.plan:
        keys = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',
                'HSA_ENABLE_IPC_MODE_LEGACY')
        child_argv = [a for a in argv if a != '--plan']
        print(json.dumps(dict(launcher='self' if 'WORLD_SIZE' not in os.environ else 'external (torch.distributed.run)',
                              backend=args.backend, collective='RCCL over xGMI' if args.backend == 'nccl' else 'gloo',
                              scaling=args.scaling,
                              rows_per_rank=[(lambda sl: sl.stop - sl.start)(_shard(args.batch, r, max(1, args.gpus)))
                                             if args.scaling == 'strong' else args.batch for r in range(max(1, args.gpus))],
                              global_batch=args.batch if args.scaling == 'strong' else args.batch * max(1, args.gpus),
                              ranks=[dict(cmd=cmd, env={k: env[k] for k in keys if k in env})
                                     for cmd, env in launch_plan(max(1, args.gpus), child_argv)]), indent=1))
        return
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(run_launcher(args, argv))

    import numpy as np
    import torch
    # stdout carries the ONE JSON line and nothing else: whatever the modules print on the way (the reference's own
    # "Using GloVe embedding", model.py:57) goes to stderr -- and so does what NATIVE code writes to file descriptor 1
    # (RCCL's version banner, gloo's rank messages): the descriptor itself is pointed at stderr, the line goes to a
    # duplicate of the original
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = 0 if args.share_gpu else int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    group = None
    forced = args.force_collectives and world == 1
    if forced:
        # one rank, real collectives: the RCCL path (library load, init with a device id, async all-reduces launched
        # from the backward, wait before Adam) runs on this GPU; no scaling information in it
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        from speaker_follower_amd import dp as _dp
        _dp.FORCE_COLLECTIVES = True
    coll = world > 1 or forced
    if coll:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group('gloo')
        group = dist.group.WORLD

    from speaker_follower_amd import synth, features, follower, _lib
    enc, dec, enc_w, dec_w = build_models(101, device)
    if args.share_gpu:
        # (test mode) several PROCESSES on one GPU: the persistent encoder launch takes every CU and its device-wide
        # lock is per process, so two of them could starve each other into the timeout path -- per-step kernels here
        enc.persistent = False
    B, S = args.batch, args.decode_steps
    table = device_table(args.n_viewpoints, 1234, device)
    store = features.FeatureStore(table, device=device)
    from speaker_follower_amd import dp as _dp_rows
    strong = args.scaling == 'strong'
    if strong:
        # ONE global batch (the same on every rank), this rank's contiguous rows of it; dropout / sampling streams are
        # keyed on the GLOBAL row id, so the sharded job draws what the unsharded batch would
        fb = synth.follower_batch(seed=0, batch=B, steps=S, n_viewpoints=args.n_viewpoints)
        my_rows = _dp_rows.shard_rows(B, rank, world)
        batch = follower.DeviceFollowerBatch.from_synth(fb, device=device, rows=my_rows, row0=my_rows.start)
    else:
        fb = synth.follower_batch(seed=rank, batch=B, steps=S, n_viewpoints=args.n_viewpoints)
        batch = follower.DeviceFollowerBatch.from_synth(fb, device=device, row0=rank * B)
    train = args.workload == 'train'
    engine = follower.FollowerEngine(enc, dec, store, group=group if train else None)
    engine.two_stream_forward = args.two_stream_forward
    if train:
        enc.train()
        dec.train()
        params_e = [p for p in enc.parameters() if p.requires_grad]
        params_d = [p for p in dec.parameters() if p.requires_grad]
        from speaker_follower_amd import dp, optim
        flat = dp.BucketedGrads(dp.follower_buckets(enc, dec), group=group)   # kernels accumulate straight into this buffer
        engine.grad_sync = flat if coll else None
        opt_e = optim.FusedAdam(params_e, lr=1e-4, weight_decay=5e-4)      # train.py:263-268
        opt_d = optim.FusedAdam(params_d, lr=1e-4, weight_decay=5e-4)
    else:
        enc.eval()
        dec.eval()
    replay = graph_state = shard_states = None
    if not train and not args.no_graph:
        # the whole episode (encoder + S decode steps + glue + loss) as ONE hipGraph: no host work per step
        if args.row_shards > 1:
            from speaker_follower_amd import dp
            shards = [follower.DeviceFollowerBatch.from_synth(
                fb, device=device, rows=dp.shard_rows(B, i, args.row_shards),
                row0=rank * B + dp.shard_rows(B, i, args.row_shards).start)
                for i in range(args.row_shards)]
            replay, shard_states, loss_buf = engine.capture_sharded(shards, S, 'argmax')

            class _Joined:                      # view of the shard results in batch order
                pass
            graph_state = _Joined()
            graph_state.loss_buf = loss_buf
        else:
            replay, graph_state = engine.capture(batch, S, 'argmax')

    def one_step():
        if train:
            flat.zero()
            st = engine.rollout(batch, S, 'argmax', train=True)
            st.loss.backward()                         # (launches the bucketed RCCL all-reduces at N > 1)
            if coll:
                flat.wait()
            opt_e.step()
            opt_d.step()
        elif replay is not None:
            replay()
            st = graph_state
        else:
            with torch.no_grad():
                st = engine.rollout(batch, S, 'argmax', train=False)
        return st

    def barrier():
        if coll:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    gc.collect()                # (BEFORE the warm-up: a collection is host time during which the device clocks fall)
    gc.disable()                # (as timeit does: no collector pause of the host inside the timed steps)
    for _ in range(args.warmup):
        st = one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt)
    agent_steps = B * S * (1 if strong else world) * args.steps
    value = agent_steps / elapsed
    if shard_states is not None:                # outside the timed region: stitch shard results
        graph_state.actions = torch.cat([x.actions for x in shard_states], dim=1)
    extras = not args.no_extras

    # ---- extra (not `value`), N > 1: the data-parallel TRAINING iteration (BASELINE configs[3]) on the
    # same per-GPU batch, with the gradient all-reduce timed on its own.  Every rank takes part.
    train_dp = None
    # The headline is measured; what follows at N > 1 issues collectives from every rank.  If that part raises on one
    # rank or hangs (a rank lost, a collective that never completes), the line with the headline is still printed: a
    # watchdog on every rank gives the extras a budget, then rank 0 prints the line it has and every rank leaves.
    headline = dict(metric='agent-steps/sec (follower rollout, batch %d)' % B, value=value, unit='agent-steps/s',
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps,
                    higher_is_better=True, scaling=args.scaling, vs_baseline=None, dtype='f32', data='synthetic',
                    config=dict(workload='follower rollout: batch %d per GPU, %d decode steps, encoder included'
                                         % (B, S)))
    watchdog = None
    if world > 1:
        import threading

        def bail():
            if rank == 0:
                headline['train_dp'] = dict(error='the data-parallel training extra did not finish within %d s; the '
                                                  'headline above was measured before it' % args.extras_budget)
                print_result(headline, result_out, args.extras_out)
            os._exit(4)
        watchdog = threading.Timer(args.extras_budget, bail)
        watchdog.daemon = True
        watchdog.start()
    if (extras or forced) and coll and not train and not args.no_train_extra:
      try:
        n_it = max(3, args.steps // 4)
        if strong:
            train_dp = measure_train(enc, dec, store, batch, S, n_it, 2, group=group, world=world, coll=coll, global_rows=B)
        else:
            train_dp = measure_train(enc, dec, store, batch, S, n_it, 2, group=group, world=world, coll=coll)
            # ... and the reference's own semantics next to it: ONE global batch of B rows split over the ranks
            fb_g = synth.follower_batch(seed=0, batch=B, steps=S, n_viewpoints=args.n_viewpoints)
            rws = _dp_rows.shard_rows(B, rank, world)
            batch_g = follower.DeviceFollowerBatch.from_synth(fb_g, device=device, rows=rws, row0=rws.start)
            train_dp['strong'] = measure_train(enc, dec, store, batch_g, S, n_it, 2, group=group, world=world, coll=coll,
                                               global_rows=B)
        if forced:
            train_dp['forced_collectives'] = ('TEST RUN: one rank, every collective of the data-parallel path issued anyway '
                                              '(%s); proves the path executes, says nothing about scaling' % args.backend)
      except Exception as exc:                      # (reported in the line; the other ranks run into the watchdog)
        import traceback
        train_dp = dict(error=repr(exc)[:400], where=traceback.format_exc()[-600:])
    if watchdog is not None:
        watchdog.cancel()
    dp_failed = bool(train_dp and train_dp.get('error'))

    # ---- extra (not `value`): serving-style throughput with several independent rollouts in flight.
    concurrent = None
    if extras and not train and replay is not None and args.in_flight > 1 and rank == 0 and world == 1:
        streams = [torch.cuda.Stream() for _ in range(args.in_flight)]
        reps = []
        for i, s in enumerate(streams):
            fbi = synth.follower_batch(seed=1000 + i, batch=B, steps=S, n_viewpoints=args.n_viewpoints)
            bi = follower.DeviceFollowerBatch.from_synth(fbi, device=device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                reps.append(follower.FollowerEngine(enc, dec, store).capture(bi, S, 'argmax') + (bi,))
        torch.cuda.synchronize()
        kk = max(args.steps, 2 * args.in_flight)
        for rnd in range(2):                                  # round 0 = warm-up
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(kk):
                with torch.cuda.stream(streams[k % args.in_flight]):
                    reps[k % args.in_flight][0]()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
        concurrent = dict(rollouts_in_flight=args.in_flight, value=B * S * kk / dt,
                          unit='agent-steps/s', ms_per_rollout=1e3 * dt / kk)

    if rank != 0:
        if coll and not dp_failed:
            torch.distributed.destroy_process_group()
        if dp_failed:
            os._exit(5)
        return

    # ---- roofline: every kernel of the rollout timed IN THIS RUN.  Three eager rollouts are issued
    # with a start/stop event pair on each dispatch (sf_profile_begin/end: the kernel's own execution
    # time on its launch stream, what rocprofv3 --kernel-trace reports); the table prices each kernel's
    # algorithmic FLOPs and bytes per launch against the fp32-MFMA and HBM peaks, `roofline` is the
    # kernel with the largest share of the rollout.
    d = synth.FULL
    H, F = d.hidden, d.feat
    D = dec.visual_attention_layer.linear_in_h.weight.shape[0]
    T = max(batch.lengths)
    A_mean = float(np.mean(fb.a_num))
    n_prof = 3
    enc.eval()
    dec.eval()
    prof_engine = follower.FollowerEngine(enc, dec, store)
    with torch.no_grad():
        prof_engine.rollout(batch, S, 'argmax', train=False)
        torch.cuda.synchronize()
        sessions = []
        for _ in range(n_prof):                       # one timing session per rollout
            with _lib.kernel_profile() as pk:
                prof_engine.rollout(batch, S, 'argmax', train=False)
            sessions.append(pk.rows)

    class prof:                                       # per kernel: the MEDIAN rollout (a one-off stall -- clock ramp, a
        rows = {}                                     # first-touch allocation -- in one rollout must not colour the table)
    for name in sessions[0]:
        per = sorted((s_[name]['avg_us'] for s_ in sessions if name in s_))
        calls = sessions[0][name]['calls']
        med = per[len(per) // 2]
        prof.rows[name] = dict(calls=calls * n_prof, avg_us=med, total_us=med * calls * n_prof,
                               min_us=min(s_[name]['min_us'] for s_ in sessions if name in s_),
                               max_us=max(s_[name]['max_us'] for s_ in sessions if name in s_))
    pmc = {}
    pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(pmc_path) and B == 100:
        pmc = json.load(open(pmc_path))['hbm_bytes_per_launch']
    busy, ghz = mfma_counters('rollout') if B == 100 else ({}, 2.4)
    kernels, kernel_us = roofline_table(prof.rows, n_prof, B, S, T, A_mean, (H, F, D, 36), pmc, busy, ghz)
    priced = [k for k in kernels if 'tflops' in k]
    top = priced[0]
    bound = 'mfma' if top['mfma_frac'] >= top['hbm_frac'] else 'hbm'
    ms_rollout = 1e3 * elapsed / args.steps
    roofline = dict(
        bound=bound,
        achieved=top['tflops'] if bound == 'mfma' else top['hbm_gbs'],
        peak=PEAK_TFLOPS_F32_MFMA if bound == 'mfma' else PEAK_HBM_GBS,
        unit='TFLOP/s' if bound == 'mfma' else 'GB/s',
        frac=top['mfma_frac'] if bound == 'mfma' else top['hbm_frac'],
        traffic=top.get('traffic_offline_pmc'),
        kernel='%s (%s): top kernel by time of the profiled rollout, %.1f%% of its kernel time'
               % (top['kernel'], top['what'], 100 * top['share']),
        launch_us=top['avg_us'], flops_per_launch=top['flops_per_launch'],
        bytes_per_launch=top['bytes_per_launch'],
        mfma_frac=top['mfma_frac'], hbm_frac=top['hbm_frac'], mfma_busy_frac=top.get('mfma_busy_frac'),
        executed=top.get('executed'),
        kernels=kernels[:8],
        kernel_time_ms_per_rollout=1e-3 * kernel_us / n_prof,
        rollout=dict(flops_frac=AGENT_STEP_FLOPS * B * S / (ms_rollout * 1e-3) / 1e12 / PEAK_TFLOPS_F32_MFMA,
                     executed_flops_frac=executed_agent_step_flops(H, F, D, 36, T, A_mean) * B * S / (ms_rollout * 1e-3)
                     / 1e12 / PEAK_TFLOPS_F32_MFMA,
                     executed_mflop_per_agent_step=1e-6 * executed_agent_step_flops(H, F, D, 36, T, A_mean),
                     hbm_frac=AGENT_STEP_BYTES * B * S / (ms_rollout * 1e-3) / 1e9 / PEAK_HBM_GBS,
                     note='flops_frac prices SURVEY 8(d)\'s per-agent-step work (71.39 MFLOP, 1.049 MB at B=100: the '
                          'UNFOLDED reference arithmetic, decode steps only) x agent-steps / ms_per_step; '
                          'executed_flops_frac prices what the kernels actually execute after the key-projection and '
                          'action-projection folds (the honest matrix-core utilisation of the whole rollout)'),
        method='launch times: HIP start/stop events on each dispatch of %d eager rollouts in this run (per kernel the median rollout); '
               'traffic_offline_pmc: HBM bytes per launch from committed rocprofv3 --pmc passes '
               '(profiles/pmc_traffic.json), NOT measured in this run' % n_prof)

    out = dict(metric='agent-steps/sec (follower rollout, batch %d)' % B, value=value,
               unit='agent-steps/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling=args.scaling,
               vs_baseline=None, dtype='f32', data='synthetic',
               config=dict(workload='follower %s: batch %d %s, 36 views x 2048-d features from a '
                                    '%d-viewpoint HBM table, <=80-token instructions, %d decode steps, '
                                    'argmax (student-forcing) feedback, every step executed for every row (no early exit), encoder included%s'
                                    % (args.workload, B, 'GLOBALLY, split over the ranks' if strong else 'per GPU',
                                       args.n_viewpoints, S,
                                       ', batch run as %d concurrent row shards' % args.row_shards
                                       if shard_states else ''),
                           global_batch=B if strong else B * world, parallelism='dp%d' % world),
               roofline=roofline, concurrent=concurrent, loss=float(st.loss_buf),
               launch=('hipGraph replay, %d concurrent row shards' % args.row_shards if shard_states
                       else 'hipGraph replay') if replay else 'eager')
    if args.share_gpu:
        out['oversubscribed'] = 'TEST RUN: %d ranks share one GPU, %s collectives' % (world, args.backend)
    if train_dp is not None:
        out['train_dp'] = train_dp

    # ---- extras (not `value`), N = 1: the other BASELINE configs on this GPU
    if extras and not train and world == 1:
        from speaker_follower_amd import bench_extras
        out['speaker_decode'] = bench_extras.speaker_decode(store, device)          # configs[2], one batch
        out['speaker_sweep'] = bench_extras.speaker_sweep(store, device)            # configs[2]: all 178 300 paths, measured
        out['speaker_train_iteration'] = bench_extras.speaker_train_iteration(store, device)   # a13, the speaker's half
        out['speaker_parity_g9'] = bench_extras.speaker_parity_g9(device)           # the hard parity set, both distances
        out['search_step'] = bench_extras.search_step(enc, dec, store, device)      # configs[4]
        conn = os.path.join(ROOT, 'tests', 'golden', 'connectivity')
        if os.path.isdir(conn):
            out['search_full'] = bench_extras.search_full(conn, device)
            out['real_env_rollout'] = bench_extras.real_env_rollout(conn, device)     # configs[1], 31-viewpoint fixture
        if args.n_viewpoints == 10567:             # configs[1] at its real size: 90 graphs, 10 567 viewpoints
            out['real_env_full'] = bench_extras.real_env_full(enc, dec, store, device)
            # configs[4] end to end through the agents' API: search + one-batch speaker rescoring + rational_mix
            out['pragmatic_inference'] = bench_extras.pragmatic_inference(enc, dec, store, device)
        # the full training iteration of configs[1] -- student-forcing rollout (dropout on), BPTT, two
        # Adam steps -- on the same batch (it updates the weights, so it runs last)
        if not args.no_train_extra:
            # (ten timed iterations behind five warm-up ones: the extra in front of it ends with host-bound work,
            # during which the device clocks fall)
            gc.collect()
            gc.freeze()        # (the full world built above is ~10^6 live Python objects: keep the collector off them)
            out['train_iteration'] = measure_train(enc, dec, store, batch, S, max(10, args.steps // 2), 10)
    # (the host-only baseline comes LAST: seconds of CPU work let the GPU clocks fall, and the extras above were
    # measured with a warm device)
    if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N = 1 only
        used = np.unique(fb.vp)
        rows = table[torch.from_numpy(used).to(device)].cpu().numpy()
        row_of = {int(v): i for i, v in enumerate(used)}
        cb, ref = cpu_baseline(enc_w, dec_w, fb, rows, row_of, S, args.cpu_reps, 1)
        out['cpu_baseline'] = cb
        if extras:
            # the same port with the BLAS pool on several host threads: best of a few pool sizes (at
            # batch 100 the products are small; every core of a 256-thread host is SLOWER than one)
            n_cpu = os.cpu_count() or 1
            tries = sorted({min(n_cpu, t) for t in (4, 16, n_cpu)})
            multi = [cpu_baseline(enc_w, dec_w, fb, rows, row_of, S, 1, t)[0] for t in tries if t > 1]
            if multi:
                best = max(multi, key=lambda c: c['value'])
                best['tried_threads'] = {str(c['cores']): c['value'] for c in multi}
                best['host_cores'] = n_cpu
                out['cpu_baseline_all_cores'] = best
        if not train:
            n = len(ref['logits'])
            same = bool(np.array_equal(st.actions.cpu().numpy()[:n], ref['actions']))
            lg = st.logits.cpu().numpy()
            worst = 0.0
            for t_ in range(n):
                a_ = ref['logits'][t_].shape[1]
                fin = np.isfinite(ref['logits'][t_])
                worst = max(worst, float(np.abs(lg[t_][:, :a_][fin] - ref['logits'][t_][fin]).max()))
            out['parity_vs_cpu_port'] = dict(actions_bit_exact=same, loss_gpu=float(st.loss_buf), loss_cpu_port=float(ref['loss']),
                                             loss_abs_diff=abs(float(st.loss_buf) - float(ref['loss'])),
                                             max_abs_logit_diff=worst, max_abs_logit=float(np.abs(lg[np.isfinite(lg)]).max()),
                                             weights='synth.follower_weights_peaky (logit std ~1.4)')
    # a starved persistent launch would have poisoned a rollout with NaN: the fault words say so (0 = healthy)
    from speaker_follower_amd import runtime as _rt
    out['persistent_launch_faults'] = _rt.take_fault(device)
    print_result(out, result_out, args.extras_out)
    if coll and not dp_failed:
        torch.distributed.destroy_process_group()
    # (after a failed data-parallel extra the group may be wedged on the other ranks: no tear-down that could block)
    # a data-parallel run in which ANY rank's persistent launch gave up a wait is not a valid measurement: say so
    # with the exit status (the line above still carries the per-rank words)
    bad = out['persistent_launch_faults'] != 0
    for td in (train_dp, (train_dp or {}).get('strong')):
        if td and td.get('health') and any(td['health']['persistent_launch_faults']):
            bad = True
    if bad:
        sys.stderr.write('bench.py: persistent-launch faults were raised (see persistent_launch_faults / train_dp.health)\n')
        sys.exit(3)
    if dp_failed:
        sys.stderr.write('bench.py: the data-parallel training extra failed (train_dp.error); the headline stands\n')
        sys.stderr.flush()
        os._exit(5)


if __name__ == '__main__':
    main()
