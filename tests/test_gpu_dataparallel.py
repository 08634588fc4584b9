"""GPU: the data-parallel path proven on ONE GPU.

The reference has no multi-GPU code; the exchange slots in between `loss.backward()` and
`optimizer.step()` (follower.py:1014-1018).  Samples interact only through the per-step loss
normaliser (CrossEntropyLoss averages over the non-ignored rows of the whole batch,
follower.py:278, 481), so R row shards + a summed [steps,2] (CE sum, live count) table + summed
gradients must reproduce the unsharded batch -- including dropout masks and sampled actions, which
are functions of the GLOBAL row id."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from speaker_follower_amd import synth                                # noqa: E402


def fresh_modules(seed=101):
    from speaker_follower_amd import model
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights(seed)
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
    return enc.cuda(), dec.cuda()


@pytest.mark.parametrize('n_shards,feedback,B', [(2, 'argmax', 26), (3, 'argmax', 26), (3, 'sample', 26), (2, 'teacher', 26),
                                                 # the shards of the 8-GPU runs: 3-4 rows, and BASELINE's 100 rows over 8 ranks
                                                 (8, 'argmax', 26), (8, 'sample', 100)])
def test_row_shards_reproduce_the_unsharded_batch(n_shards, feedback, B):
    from speaker_follower_amd import follower, features, dp
    S, NVP = 7, 64
    enc, dec = fresh_modules()
    enc.train()
    dec.train()                                            # dropout ON: masks must follow the global row
    params = [p for p in list(enc.parameters()) + list(dec.parameters()) if p.requires_grad]
    flat = dp.FlatGrads(params)
    fb = synth.follower_batch(seed=3, batch=B, steps=S, n_viewpoints=NVP, min_len=4, max_len=30, a_max=9)
    store = features.FeatureStore(synth.feature_table(3, NVP))

    def engine():
        e = follower.FollowerEngine(enc, dec, store)
        e.dropout_seed = 0x1234ABCD
        return e

    whole = follower.DeviceFollowerBatch.from_synth(fb)
    st = engine().rollout(whole, S, feedback, train=True)
    st.loss.backward()
    torch.cuda.synchronize()
    ref_loss = float(st.loss.detach())
    ref_grad = flat.flat.clone()
    ref_actions = st.actions.cpu().numpy()
    ref_cnt = st.sum_cnt.cpu().numpy()
    assert np.isfinite(ref_loss) and float(ref_grad.abs().max()) > 0

    flat.zero()
    shards, states = [], []
    for i in range(n_shards):
        rows = dp.shard_rows(B, i, n_shards)
        sh = follower.DeviceFollowerBatch.from_synth(fb, rows=rows, row0=rows.start)
        shards.append(rows)
        states.append((engine(), sh))
    states = [(e, e.rollout(sh, S, feedback, train=True, finalize=False)) for e, sh in states]
    total = sum(s.sum_cnt for _, s in states)              # what the [steps,2] all-reduce produces
    np.testing.assert_allclose(total.cpu().numpy(), ref_cnt, rtol=1e-6, atol=1e-6)
    for e, s in states:
        e.finish(s, total.clone())
        assert abs(float(s.loss.detach()) - ref_loss) <= 1e-6 * max(1.0, abs(ref_loss))
        s.loss.backward()                                  # accumulates into the one flat buffer
    torch.cuda.synchronize()
    got_actions = np.concatenate([s.actions.cpu().numpy() for _, s in states], axis=1)
    assert np.array_equal(got_actions, ref_actions)        # same masks, same samples, same argmax
    got, want = flat.flat, ref_grad
    gmax = float(want.abs().max())
    off = 0
    for p in params:         # per parameter: 2e-6 of its own scale (+ 1e-7 of the largest gradient: the
        n = p.numel()        # visual linear_in_v.bias and scoring linear_out.bias gradients are exactly 0 in exact arithmetic)
        g, w = got[off:off + n], want[off:off + n]
        scale = float(w.abs().max())
        assert float((g - w).abs().max()) <= 2e-6 * scale + 1e-7 * gmax, (tuple(p.shape), scale, gmax)
        off += n


def test_bucketed_gradient_schedule_gives_the_same_gradients():
    """dp.BucketedGrads hooked into the backward (FollowerEngine.grad_sync): the weight-gradient call split into
    its LSTM part and the rest, each followed by its bucket's launch, then the encoder's -- every gradient equal to
    the plain schedule's, the three buckets launched in production order, wait() accepted; a rollout whose
    backward never ran is refused at wait()."""
    from speaker_follower_amd import follower, features, dp
    B, S, NVP = 20, 6, 64
    enc, dec = fresh_modules(11)
    enc.train()
    dec.train()
    fb = synth.follower_batch(seed=4, batch=B, steps=S, n_viewpoints=NVP, min_len=4, max_len=30, a_max=9)
    store = features.FeatureStore(synth.feature_table(4, NVP))
    batch = follower.DeviceFollowerBatch.from_synth(fb)

    def run(sync):
        eng = follower.FollowerEngine(enc, dec, store)
        eng.dropout_seed = 777
        eng.grad_sync = sync
        st = eng.rollout(batch, S, 'teacher', train=True)
        st.loss.backward()
        torch.cuda.synchronize()
        return float(st.loss)

    plain = dp.FlatGrads([p for p in list(enc.parameters()) + list(dec.parameters()) if p.requires_grad])
    plain.zero()
    loss_a = run(None)
    ref = {id(p): p.grad.clone() for p in plain.params}
    buckets = dp.follower_buckets(enc, dec)
    sync = dp.BucketedGrads(buckets)                    # re-points every .grad into the bucketed buffer
    assert sync.n_buckets == 3 and [hi - lo for lo, hi in sync.bounds] == [2048 * 4864 + 4096, 12129537 - 2048 * 4864 - 4096, 1929728]
    sync.zero()
    loss_b = run(sync)
    assert sync.launched == [0, 1, 2]                   # decoder LSTM, other decoder weights, encoder
    sync.wait()
    assert loss_a == loss_b
    for p in sync.params:
        scale = float(ref[id(p)].abs().max())
        assert float((p.grad - ref[id(p)]).abs().max()) <= 1e-6 * max(scale, 1e-6)
    # the decoder LSTM bucket is the head of the buffer
    assert dec.lstm.weight_ih.grad.data_ptr() == sync.flat.data_ptr()
    with pytest.raises(RuntimeError, match='never launched'):
        sync.launch(0)
        sync.wait()


def test_captured_rollout_follows_weight_updates():
    """A hipGraph replay after optimizer.step() must use the NEW weights everywhere -- also through
    the cached transposed copies and the encoder's embedding x W_ih^T table (rebuilt in place)."""
    from speaker_follower_amd import follower, features, runtime
    B, S, NVP = 10, 4, 32
    enc, dec = fresh_modules(7)
    enc.eval()
    dec.eval()
    fb = synth.follower_batch(seed=5, batch=B, steps=S, n_viewpoints=NVP, min_len=4, max_len=20, a_max=8)
    store = features.FeatureStore(synth.feature_table(5, NVP))
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    eng = follower.FollowerEngine(enc, dec, store)
    replay, gst = eng.capture(batch, S, 'argmax')

    def eager():
        with torch.no_grad():
            return follower.FollowerEngine(enc, dec, store).rollout(batch, S, 'argmax', train=False)

    replay()
    torch.cuda.synchronize()
    before = gst.logits.clone()
    assert torch.equal(before, eager().logits)
    with torch.no_grad():                                   # an update torch knows about (bumps _version)
        for p in list(enc.parameters()) + list(dec.parameters()):
            if p.requires_grad:
                p.mul_(1.25)
    replay()
    torch.cuda.synchronize()
    after = gst.logits.clone()
    fin = torch.isfinite(before)
    assert not torch.allclose(after[fin], before[fin])
    assert torch.equal(after, eager().logits)
    # an update torch does NOT see (`.data` has its own version counter) needs the explicit call
    for p in dec.parameters():
        p.data.mul_(0.8)
    for p in enc.parameters():
        if p.requires_grad:
            p.data.mul_(0.8)
    runtime.invalidate_caches()
    replay()
    torch.cuda.synchronize()
    assert torch.equal(gst.logits, eager().logits)
    assert not torch.allclose(gst.logits[fin], after[fin])
    # a re-allocated weight cannot be patched into the graph: loud failure, not stale memory
    dec.lstm.weight_hh.data = dec.lstm.weight_hh.data.clone()
    with pytest.raises(RuntimeError, match='capture'):
        replay()


def test_dropout_sites_do_not_collide_for_long_rollouts():
    """Sites advance by at least S + 2 per rollout (they were `iteration * 64`: a rollout with more
    than 62 steps reused the masks of the next iteration)."""
    from speaker_follower_amd import follower, features
    enc, dec = fresh_modules(9)
    store = features.FeatureStore(synth.feature_table(1, 16))
    eng = follower.FollowerEngine(enc, dec, store)
    fb = synth.follower_batch(seed=1, batch=4, steps=70, n_viewpoints=16, min_len=3, max_len=8, a_max=5)
    batch = follower.DeviceFollowerBatch.from_synth(fb)
    with torch.no_grad():
        a = eng.rollout(batch, 70, 'argmax', train=True)
        b = eng.rollout(batch, 5, 'argmax', train=True)
        c = eng.rollout(batch, 5, 'argmax', train=True)
    assert a.site0 == 0 and b.site0 == 72 and c.site0 == 72 + 64


def test_bench_two_ranks_on_one_gpu_over_gloo(tmp_path):
    """The N > 1 code path of bench.py end to end -- self-launch, rendezvous, weak-scaling rollout,
    data-parallel training iteration with the flat gradient all-reduce, one JSON line from rank 0 --
    with both ranks sharing this box's single GPU and gloo standing in for RCCL."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--share-gpu',
                          '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--n-viewpoints', '96',
                          '--batch', '16', '--decode-steps', '5', '--extras-out', str(tmp_path / 'x.json')],
                         capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 32 and d['config']['parallelism'] == 'dp2'
    assert d['scaling'] == 'weak' and 'oversubscribed' in d
    assert abs(d['value'] - 2 * 16 * 5 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']
    # the line is compact (the driver holds a few KB): scalars of the data-parallel iteration only; the full object is
    # in the extras file
    assert len(lines[0]) < 4096 and d['train_dp']['allreduce_ms'] > 0 and d['train_dp']['faults'] == 0
    assert d['train_dp']['strong']['global_batch'] == 16
    t = json.load(open(tmp_path / 'x.json'))['train_dp']
    assert t['allreduce_ms'] > 0 and t['allreduce_bytes'] == 4 * 14059265 and np.isfinite(t['loss'])
    # the bucketed schedule: decoder LSTM (weight_ih, weight_hh, two biases), other decoder weights, encoder
    assert t['buckets_bytes'] == [4 * (2048 * 4352 + 2048 * 512 + 4096), 4 * (12129537 - 2048 * 4864 - 4096), 4 * 1929728]
    assert t['allreduce_exposed_ms_overlapped'] >= 0 and t['ms_per_iteration_no_exchange'] > 0
    # per-rank health of the persistent launches + the production entry point (FollowerEngine.run) under the group
    assert t['scaling'] == 'weak' and t['health']['persistent_launch_faults'] == [0, 0] and t['health']['fallbacks'] == [0, 0]
    # the reference's own semantics next to it: ONE global batch split over the ranks
    ts = t['strong']
    assert ts['scaling'] == 'strong' and ts['global_batch'] == 16 and ts['rows_this_rank'] == 8
    assert ts['health']['persistent_launch_faults'] == [0, 0] and np.isfinite(ts['loss'])


def test_bench_strong_scaling_two_ranks_on_one_gpu_over_gloo(tmp_path):
    """`--scaling strong`: the headline loop itself over ONE global batch split with dp.shard_rows."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--share-gpu',
                          '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--n-viewpoints', '96',
                          '--batch', '18', '--decode-steps', '5', '--scaling', 'strong', '--no-cpu-baseline',
                          '--extras-out', str(tmp_path / 'x.json')],
                         capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')})
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['scaling'] == 'strong' and d['n_gpus'] == 2 and d['config']['global_batch'] == 18
    assert abs(d['value'] - 18 * 5 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']
    assert d['train_dp']['scaling'] == 'strong' and d['train_dp']['global_batch'] == 18
    t = json.load(open(tmp_path / 'x.json'))['train_dp']
    assert t['scaling'] == 'strong' and t['global_batch'] == 18 and t['rows_this_rank'] == 9 and 'strong' not in t


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('backend,world,feedback', [('gloo', 2, 'sample'), ('gloo', 2, 'teacher'), ('nccl', 1, 'sample')])
def test_data_parallel_iteration_as_graph_segments(backend, world, feedback):
    """a process group no longer forces the training iteration back to launch-by-launch issue.
    FollowerEngine.capture_training under a group captures the iteration as SEGMENTS cut at the collective points
    (count-table all-reduce, the three gradient buckets, the wait): per rank the replayed iterations give the eager
    data-parallel loop's losses, sampled actions, Adam steps and weights; every rank ends with the same weights."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
        env.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp_graph_worker.py'), backend, feedback],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, so[-1500:] + se[-4000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith('DP_GRAPH_WORKER ')][-1][len('DP_GRAPH_WORKER '):]))
    for o in outs:
        print({k: o[k] for k in ('rank', 'rows', 'segments', 'losses_eager', 'losses_graph', 'weight_rel_diff')})
        assert o['segments'] == 6                                   # counts | bucket 0 | bucket 1 | bucket 2 | wait | (Adam)
        assert o['actions_equal'] and o['steps'][0] == o['steps'][1] == [4, 4]
        np.testing.assert_allclose(o['losses_graph'], o['losses_eager'], rtol=2e-6)
        assert len(set(o['losses_eager'])) == 4 and o['weights_finite'] and o['moved'] > 0
        assert o['weight_rel_diff'] <= 2e-6                          # (two-stream backward: accumulation order across streams)
    sums = outs[0]['weight_sums_by_rank']
    assert all(abs(s - sums[0]) <= 1e-9 * abs(sums[0]) for s in sums)       # replicas stay replicas
    if world > 1:
        assert outs[0]['losses_eager'] == outs[1]['losses_eager']           # every rank holds the GLOBAL loss


def test_agent_train_under_a_process_group_replays_segments():
    """agents.Seq2SeqAgent.train with an engine that carries a process group and gradient buckets no longer falls back to
    launch-by-launch issue (agents._graph_trainable): iterations replay as segments, with the fault word reduced over the
    group before any rank re-issues.  One rank, every collective issued: the numbers of the plain graph loop."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_agent_worker.py')], env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-4000:]
    o = json.loads([l for l in res.stdout.splitlines() if l.startswith('DP_AGENT_WORKER ')][-1][len('DP_AGENT_WORKER '):])
    print(o)
    assert o['segments'] == [0, 6] and o['replays'] == [4, 4] and o['fallbacks'] == [0, 0] and o['finite']
    np.testing.assert_allclose(o['losses_group'], o['losses_plain'], rtol=2e-6)
    assert len(set(o['losses_plain'])) == 5 and o['weight_rel_diff'] <= 2e-6
