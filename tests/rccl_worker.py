"""Worker of tests/test_gpu_rccl.py (its own process: it initialises a process group).

ONE rank, real collectives: `init_process_group('nccl', device_id=...)` on the GPU, then the data-parallel training
iteration of bench.py -- count-table all-reduce inside the rollout, the gradient buckets' async all-reduces launched
from the backward, wait(), FusedAdam -- next to the same iteration with no process group at all.  RCCL's all-reduce
over one rank is the identity, so loss, every gradient and every updated weight must be BIT-identical; what the run
proves is that the library loads, the environment is right and the stream ordering of launch / wait holds under NCCL
semantics (the collective runs on the process group's stream, ordered by events against the launching stream).
Prints one JSON line."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(backend):
    import torch
    import torch.distributed as dist
    from speaker_follower_amd import synth, features, follower, dp, optim, model
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    d = synth.FULL
    B, S, NVP = 48, 6, 128
    fb = synth.follower_batch(seed=2, batch=B, steps=S, n_viewpoints=NVP)
    store = features.FeatureStore(synth.feature_table(5, NVP), device=dev)
    batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)

    def build():
        enc_w, dec_w = synth.follower_weights_peaky(31)
        enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
        dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
        enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
        dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
        return enc.to(dev).train(), dec.to(dev).train()

    def iterations(group, n=3):
        enc, dec = build()
        flat = dp.BucketedGrads(dp.follower_buckets(enc, dec), group=group)
        opt_e = optim.FusedAdam([p for p in enc.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
        opt_d = optim.FusedAdam([p for p in dec.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
        eng = follower.FollowerEngine(enc, dec, store, group=group)
        eng.dropout_seed = 99
        eng.grad_sync = flat if group is not None else None
        losses, grads = [], None
        for _ in range(n):
            flat.zero()
            st = eng.rollout(batch, S, 'sample', train=True)
            st.loss.backward()
            if group is not None:
                flat.wait()
            grads = flat.flat.clone()
            opt_e.step()
            opt_d.step()
            losses.append(float(st.loss.detach()))
        torch.cuda.synchronize()
        weights = torch.cat([p.detach().reshape(-1) for p in list(enc.parameters()) + list(dec.parameters())])
        return losses, grads, weights, flat

    base = iterations(None)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=dev)
    else:
        dist.init_process_group('gloo')
    dp.FORCE_COLLECTIVES = True
    got = iterations(dist.group.WORLD)
    # the blocking form and the count-table helper too
    t = torch.arange(12, dtype=torch.float32, device=dev).reshape(6, 2).contiguous()
    want = t.clone()
    dp.allreduce_step_counts(t, dist.group.WORLD)
    got[3].allreduce(dist.group.WORLD)
    torch.cuda.synchronize()

    # FollowerEngine.run(backward=True) under the process group with a persistent launch that gives up its wait (forced):
    # the fault flag is reduced over the group (MAX: RCCL has no bitwise reductions), the gradient buckets the
    # poisoned backward launched are waited for and re-armed, and the re-issued iteration leaves the gradients of an
    # undisturbed one.
    from speaker_follower_amd import _lib, runtime

    def one_run(faulty):
        enc, dec = build()
        flat = dp.BucketedGrads(dp.follower_buckets(enc, dec), group=dist.group.WORLD)
        eng = follower.FollowerEngine(enc, dec, store, group=dist.group.WORLD)
        eng.dropout_seed = 99
        eng.grad_sync = flat
        runtime.take_fault(dev)
        _lib.lib.sf_debug_persist_timeout(0 if faulty else -1)
        try:
            st = eng.run(batch, S, 'teacher', train=True, backward=True)
            flat.wait()
        finally:
            _lib.lib.sf_debug_persist_timeout(-1)
        torch.cuda.synchronize()
        runtime.take_fault(dev)
        return eng.fallbacks, float(st.loss.detach()), flat.flat.clone()

    clean, faulted = one_run(False), one_run(True)
    gscale = float(clean[2].abs().max())
    fault = dict(fallbacks=[clean[0], faulted[0]], losses=[clean[1], faulted[1]],
                 finite=bool(torch.isfinite(faulted[2]).all()),
                 grad_rel_diff=float((clean[2] - faulted[2]).abs().max()) / max(gscale, 1e-12))
    out = dict(fault=fault, backend=dist.get_backend(), world=dist.get_world_size(),
               nccl_version='.'.join(map(str, torch.cuda.nccl.version())) if backend == 'nccl' else None,
               losses_equal=base[0] == got[0], losses=got[0],
               grads_bit_identical=bool(torch.equal(base[1], got[1])),
               weights_bit_identical=bool(torch.equal(base[2], got[2])),
               grad_abs_max=float(got[1].abs().max()), finite=bool(torch.isfinite(got[1]).all()),
               counts_identity=bool(torch.equal(t, want)), buckets=got[3].n_buckets,
               bucket_bytes=[4 * (hi - lo) for lo, hi in got[3].bounds])
    dist.destroy_process_group()
    print('RCCL_WORKER ' + json.dumps(out))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'nccl')
