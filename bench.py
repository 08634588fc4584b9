#!/usr/bin/env python3
"""Headline benchmark: follower rollout throughput in agent-steps/s (BASELINE.json).

One "step" of this script = one follower episode batch on every rank: EncoderLSTM over the
(<=80-token) instructions + `--decode-steps` AttnDecoderLSTM steps with on-device argmax
feedback, masking, cross-entropy and u_prev gather, over index-form observations gathered
from the HBM-resident 36x2048 feature table.  With --workload train the step also runs
BPTT, the (data-parallel) gradient all-reduce and two Adam updates (train.py:263-268).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  value = B * decode_steps * world * K / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_LSTM = lambda B, I, H: 2.0 * B * (I + H) * 4 * H      # noqa: E731  gate GEMM of one LSTMCell step


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', choices=['rollout', 'train'], default='rollout')
    ap.add_argument('--batch', type=int, default=100)
    ap.add_argument('--decode-steps', type=int, default=20)
    ap.add_argument('--n-viewpoints', type=int, default=10567)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-train-extra', action='store_true', help='skip the extra training-iteration measurement')
    ap.add_argument('--no-graph', action='store_true', help='issue the rollout eagerly instead of replaying a hipGraph')
    ap.add_argument('--row-shards', type=int, default=1,
                    help='rollout only: run the batch as this many concurrent row shards (measured: no gain, '
                         'a half-batch chain is as long as a full one)')
    ap.add_argument('--in-flight', type=int, default=2,
                    help='extra measurement: this many independent batch-100 rollouts in flight on separate streams')
    ap.add_argument('--cpu-reps', type=int, default=2)
    return ap.parse_args()


def build_models(seed, device):
    from speaker_follower_amd import synth, model
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return enc.to(device), dec.to(device), enc_w, dec_w


def device_table(n_vp, seed, device):
    """ResNet-pool5-like table generated on the device (0.5*N(0,1) clipped at 0), 3.1 GB at
    the full 10 567 viewpoints."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.empty(n_vp, 36, 2048, device=device, dtype=torch.float32)
    chunk = 512
    for i in range(0, n_vp, chunk):
        blk = t[i:i + chunk]
        blk.normal_(0.0, 0.5, generator=g)
        blk.clamp_(min=0.0)
    return t


def cpu_baseline(enc_w, dec_w, fb, table_rows, row_of, decode_steps, reps):
    """The numpy oracle (a port of the reference modules, oracle/np_model.py) on ONE host thread,
    on the same batch the GPU ran: full rollout, `reps` repetitions."""
    from threadpoolctl import threadpool_limits
    from oracle import np_env, np_model
    import copy
    fbc = copy.copy(fb)
    fbc.vp = np.vectorize(row_of.get)(fb.vp).astype(np.int32)
    loc = np_env.static_loc_embeddings()
    seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, 80, reverse=True)
    B = len(lens)
    best = None
    with threadpool_limits(limits=1):
        for _ in range(reps):
            t0 = time.perf_counter()
            res = np_model.follower_rollout(
                enc_w, dec_w, seq, lens, mask, decode_steps,
                lambda t: np_env.dense_follower_step(table_rows, loc, fbc, t), fb.target, 'argmax',
                2176, early_exit=False)       # same work as the GPU: every step for every row
            dt = time.perf_counter() - t0
            n = len(res['logits'])
            rate = B * n / dt
            best = rate if best is None else max(best, rate)
    return dict(value=best, unit='agent-steps/s', cores=1, kind='port',
                sample='%d full rollouts of the same batch (B=%d, %d decode steps, encoder included), '
                       'numpy oracle, 1 thread, best of %d' % (reps, B, n, reps)), res


def measure_train(enc, dec, store, batch, S, iters, warmup):
    """follower.py:1001-1020 + train.py:263-268 per iteration: zero_grad, student-forcing rollout with
    loss, backward, Adam(lr 1e-4, weight_decay 5e-4) on encoder and decoder."""
    from speaker_follower_amd import follower, dp
    enc.train()
    dec.train()
    params_e = [p for p in enc.parameters() if p.requires_grad]
    params_d = [p for p in dec.parameters() if p.requires_grad]
    from speaker_follower_amd import optim
    flat = dp.FlatGrads(params_e + params_d)      # gradients: one buffer (what the all-reduce wants)
    opt_e = optim.FusedAdam(params_e, lr=1e-4, weight_decay=5e-4)      # one launch per step each
    opt_d = optim.FusedAdam(params_d, lr=1e-4, weight_decay=5e-4)
    engine = follower.FollowerEngine(enc, dec, store)
    B = batch.batch_size

    def it():
        flat.zero()
        st = engine.rollout(batch, S, 'argmax', train=True)
        st.loss.backward()
        opt_e.step()
        opt_d.step()
        return st
    for _ in range(warmup):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        st = it()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    enc.eval()
    dec.eval()
    return dict(value=B * S / dt, unit='agent-steps/s', ms_per_iteration=1e3 * dt, iterations=iters,
                what='student-forcing rollout (dropout 0.5) + BPTT + 2x Adam (one HIP launch each), batch %d, %d decode steps, '
                     'eager issue' % (B, S), loss=float(st.loss.detach()))


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    group = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=device)
        group = dist.group.WORLD

    from speaker_follower_amd import synth, features, follower
    enc, dec, enc_w, dec_w = build_models(101, device)
    B, S = args.batch, args.decode_steps
    table = device_table(args.n_viewpoints, 1234, device)
    store = features.FeatureStore(table, device=device)
    fb = synth.follower_batch(seed=rank, batch=B, steps=S, n_viewpoints=args.n_viewpoints)
    batch = follower.DeviceFollowerBatch.from_synth(fb, device=device, row0=rank * B)
    train = args.workload == 'train'
    engine = follower.FollowerEngine(enc, dec, store, group=group if train else None)
    if train:
        enc.train()
        dec.train()
        params_e = [p for p in enc.parameters() if p.requires_grad]
        params_d = [p for p in dec.parameters() if p.requires_grad]
        from speaker_follower_amd import dp, optim
        flat = dp.FlatGrads(params_e + params_d)       # kernels accumulate straight into this buffer
        opt_e = optim.FusedAdam(params_e, lr=1e-4, weight_decay=5e-4)      # train.py:263-268
        opt_d = optim.FusedAdam(params_d, lr=1e-4, weight_decay=5e-4)
    else:
        enc.eval()
        dec.eval()
    replay = graph_state = shard_states = None
    if not train and not args.no_graph:
        # the whole episode (encoder + S decode steps + glue + loss) as ONE hipGraph: ~330 kernels,
        # no host work per step
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
            st.loss.backward()
            flat.allreduce(group)                      # one RCCL sum all-reduce of 56 MB (no-op at N=1)
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
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        st = one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt)
    agent_steps = B * S * world * args.steps
    value = agent_steps / elapsed
    if shard_states is not None:                # outside the timed region: stitch shard results
        graph_state.actions = torch.cat([x.actions for x in shard_states], dim=1)

    # ---- extra (not `value`): serving-style throughput with several independent rollouts in flight.
    # Every stage of one batch-100 chain is latency-bound and leaves most CUs idle, so independent
    # episodes overlap on separate streams (validation / data-augmentation decode over many batches).
    concurrent = None
    if not train and replay is not None and args.in_flight > 1 and rank == 0:
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
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    # ---- roofline of the dominant kernel: the decoder LSTMCell gate GEMM [B,4864]x[4864,2048]
    from speaker_follower_amd import ops
    H, I = 512, 4352
    x = torch.randn(B, I, device=device)
    h0 = torch.randn(B, H, device=device)
    c0 = torch.randn(B, H, device=device)
    w4 = [dec.lstm.weight_ih.detach(), dec.lstm.weight_hh.detach(), dec.lstm.bias_ih.detach(),
          dec.lstm.bias_hh.detach()]
    from speaker_follower_amd.runtime import ptr, ws_args, struct_of
    from speaker_follower_amd import _lib
    import ctypes as C
    # exactly what the rollout launches for the gates: x W_ih^T + h W_hh^T as split-K slabs (the
    # LSTM pointwise kernel adds slabs + biases); ONE kernel per call, timed with HIP events on the
    # stream it is launched on, over `reps` back-to-back launches
    ks = C.c_int(0)
    reps = 100
    for _ in range(10):
        _lib.call('sf_linear_slabs_fwd', ptr(x), I, ptr(w4[0]), I, ptr(h0), H, ptr(w4[1]), H, B, 4 * H,
                  C.byref(ks), *ws_args(device))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        _lib.call('sf_linear_slabs_fwd', ptr(x), I, ptr(w4[0]), I, ptr(h0), H, ptr(w4[1]), H, B, 4 * H,
                  C.byref(ks), *ws_args(device))
    e1.record()
    torch.cuda.synchronize()
    gemm_ms = e0.elapsed_time(e1) / reps
    flops = FLOPS_LSTM(B, I, H)
    achieved = flops / (gemm_ms * 1e-3) / 1e12
    traffic = None          # HBM bytes per launch from the committed rocprofv3 PMC passes (offline)
    pmc = os.path.join(ROOT, 'profiles', 'r01_lstm_gemm_pmc.json')
    if os.path.exists(pmc) and B == 100:
        traffic = json.load(open(pmc))['hbm_bytes_per_launch']
    roofline = dict(bound='mfma', achieved=achieved, peak=157.3, unit='TFLOP/s',
                    frac=achieved / 157.3, traffic=traffic,
                    kernel='gemm_nt_tiled_kernel<7> (decoder LSTMCell gate product '
                           '[%d,%d]x[%d,%d]^T fp32 -> %d split-K slabs, summed by lstm_pw_fwd_kernel)'
                           % (B, I + H, 4 * H, I + H, ks.value),
                    launch_ms=gemm_ms, flops_per_launch=flops,
                    note='peak = 2.4 GHz fp32 MFMA; the same MFMA stream alone sustains ~145 TFLOP/s on '
                         'random operands (tools/exp/mfma_power.hip)')

    out = dict(metric='agent-steps/sec (follower rollout, batch %d)' % B, value=value,
               unit='agent-steps/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling='weak',
               vs_baseline=None, dtype='f32', data='synthetic',
               config=dict(workload='follower %s: batch %d per GPU, 36 views x 2048-d features from a '
                                    '%d-viewpoint HBM table, <=80-token instructions, %d decode steps, '
                                    'argmax (student-forcing) feedback, every step executed for every row (no early exit), encoder included%s'
                                    % (args.workload, B, args.n_viewpoints, S,
                                       ', batch run as %d concurrent row shards' % args.row_shards
                                       if shard_states else ''),
                           global_batch=B * world, parallelism='dp%d' % world),
               roofline=roofline, concurrent=concurrent, loss=float(st.loss_buf), launch=('hipGraph replay, %d concurrent row shards' % args.row_shards if shard_states
                       else 'hipGraph replay') if replay else 'eager')

    if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N = 1 only
        used = np.unique(fb.vp)
        rows = table[torch.from_numpy(used).to(device)].cpu().numpy()
        row_of = {int(v): i for i, v in enumerate(used)}
        cb, ref = cpu_baseline(enc_w, dec_w, fb, rows, row_of, S, args.cpu_reps)
        out['cpu_baseline'] = cb
        if not train:
            n = len(ref['logits'])
            same = bool(np.array_equal(st.actions.cpu().numpy()[:n], ref['actions']))
            out['parity_vs_cpu_port'] = dict(actions_bit_exact=same,
                                             loss_abs_diff=abs(float(st.loss_buf) - float(ref['loss'])))
    # ---- extra (not `value`): the full training iteration of BASELINE configs[1] -- student-forcing
    # rollout (dropout on), BPTT through the C ABI, two Adam steps -- on the same batch, N = 1 only
    # (it updates the weights, so it runs after every inference measurement and parity check)
    if not train and world == 1 and not args.no_train_extra:
        out['train_iteration'] = measure_train(enc, dec, store, batch, S, max(3, args.steps // 4), 2)
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
