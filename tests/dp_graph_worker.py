"""Worker of tests/test_gpu_dataparallel.py::test_data_parallel_iteration_as_graph_segments (one process per rank; every
rank uses cuda:0 -- the box has one GPU -- and the collectives go through `backend`).

ONE global batch split over the ranks with dp.shard_rows; N training iterations (student-forcing rollout with dropout,
count-table all-reduce, BPTT, bucketed gradient all-reduce, two Adam steps) issued (a) eagerly, launch by launch, and
(b) as runtime.TrainingGraph segments -- the capture cut at every collective point, a replay = segment, host action,
segment, ... (FollowerEngine._capture_training_dp).  Prints one JSON line: losses and a weight checksum of both."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(backend, feedback):
    import torch
    import torch.distributed as dist
    from speaker_follower_amd import synth, features, follower, dp, optim, model
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=dev)
    else:
        dist.init_process_group('gloo')
    if world == 1:
        dp.FORCE_COLLECTIVES = True
    group = dist.group.WORLD
    d = synth.FULL
    B, S, NVP, N = 26, 5, 96, 4
    fb = synth.follower_batch(seed=2, batch=B, steps=S, n_viewpoints=NVP, min_len=5, max_len=40)
    store = features.FeatureStore(synth.feature_table(5, NVP), device=dev)
    rows = dp.shard_rows(B, rank, world)
    batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev, rows=rows, row0=rows.start)

    def build():
        enc_w, dec_w = synth.follower_weights_peaky(31)
        enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
        dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
        enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
        dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
        enc.to(dev).train()
        dec.to(dev).train()
        if world > 1:
            enc.persistent = False          # (several PROCESSES share this GPU: their persistent launches would starve each other)
        return enc, dec

    def run(mode):
        enc, dec = build()
        flat = dp.BucketedGrads(dp.follower_buckets(enc, dec), group=group)
        oe = optim.FusedAdam([p for p in enc.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-4)
        od = optim.FusedAdam([p for p in dec.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-4)
        eng = follower.FollowerEngine(enc, dec, store, group=group)
        eng.dropout_seed = 99
        eng.grad_sync = flat
        losses, acts = [], []
        if mode == 'eager':
            for _ in range(N):
                flat.zero()
                st = eng.rollout(batch, S, feedback, train=True)
                st.loss.backward()
                flat.wait()
                oe.step()
                od.step()
                losses.append(float(st.loss.detach()))
                acts.append(st.actions.cpu().numpy().tolist())
            segs = 0
        else:
            tg = eng.capture_training(batch, S, feedback, optimizers=(oe, od), zero=flat)
            losses.append(float(tg.first.loss_buf))
            acts.append(tg.first.actions.cpu().numpy().tolist())
            for _ in range(N - 1):
                st = tg.replay()
                torch.cuda.synchronize()
                losses.append(float(st.loss_buf))
                acts.append(st.actions.cpu().numpy().tolist())
            segs = len(tg.segments)
            assert flat.hook is None and eng.collective_hook is None
        torch.cuda.synchronize()
        w = torch.cat([p.detach().reshape(-1) for p in list(enc.parameters()) + list(dec.parameters())])
        return losses, acts, w, segs, oe.host_steps() + od.host_steps()

    le, ae, we, _, se = run('eager')
    lg, ag, wg, segs, sg = run('graph')
    scale = float(we.abs().max())
    every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(every, we.double().sum().reshape(1))
    out = dict(rank=rank, world=world, backend=dist.get_backend(), rows=[rows.start, rows.stop], segments=segs,
               losses_eager=le, losses_graph=lg, actions_equal=ae == ag, steps=[se, sg],
               weight_rel_diff=float((we - wg).abs().max()) / scale, weights_finite=bool(torch.isfinite(wg).all()),
               weight_sums_by_rank=[float(x) for x in every], moved=float((we - torch.cat([p.detach().reshape(-1) for p in
                                                                                         sum((list(m.parameters()) for m in build()), [])])).abs().max()))
    dist.barrier()
    dist.destroy_process_group()
    print('DP_GRAPH_WORKER ' + json.dumps(out), flush=True)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'sample')
