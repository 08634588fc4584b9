"""CPU: host-side logic of the product (no kernels): instruction batching, location table,
candidate sin/cos, synthetic batches, data-parallel helpers (gloo, world_size 2)."""
import os

import numpy as np
import pytest
import torch

from speaker_follower_amd import synth


def test_batch_instructions_matches_reference_golden(golden):
    from speaker_follower_amd.follower import batch_instructions_from_encoded
    g = golden('g6_env')
    toks = np.split(g['instr_tokens'], np.cumsum(g['instr_sizes'])[:-1])
    for tag, rev in (('fwd', False), ('rev', True)):
        seq, mask, lens = batch_instructions_from_encoded(toks, 80, reverse=rev, device='cpu')
        np.testing.assert_array_equal(seq.numpy(), g['instr_seq_' + tag])
        np.testing.assert_array_equal(mask.numpy(), g['instr_mask_' + tag])
        assert lens == list(g['instr_len_' + tag])
    seq, mask, lens, perm = batch_instructions_from_encoded(toks, 80, reverse=True, sort=True,
                                                           device='cpu')
    np.testing.assert_array_equal(seq.numpy(), g['instr_seq_sorted'])
    assert lens == list(g['instr_len_sorted'])
    assert lens == sorted(lens, reverse=True)


def test_batch_instructions_edge_cases():
    from speaker_follower_amd.follower import batch_instructions_from_encoded
    seq, mask, lens = batch_instructions_from_encoded([[], [5] * 200], 80, reverse=True, device='cpu')
    assert lens == [1, 80]                      # empty -> just EOS; long -> truncated (EOS dropped)
    assert int(seq[0, 0]) == 2 and int(seq[1, 79]) == 5
    assert mask.shape == (2, 80) and bool(mask[0, 1]) and not bool(mask[1, 79])


def test_loc_table_and_sincos_match_reference_golden(golden):
    from speaker_follower_amd import features
    g = golden('g6_env')
    np.testing.assert_array_equal(features.build_loc_table(), g['loc_table'])
    sc = features.cand_sincos(g['act_cand_heading'], g['act_cand_elevation'])
    emb = g['act_embedding']
    for a in range(1, len(sc)):
        np.testing.assert_array_equal(emb[a, 2048:2048 + 128:32], sc[a])


def test_synth_batches_are_deterministic_and_well_formed():
    a = synth.follower_batch(seed=3, batch=16, steps=12, n_viewpoints=50)
    b = synth.follower_batch(seed=3, batch=16, steps=12, n_viewpoints=50)
    np.testing.assert_array_equal(a.vp, b.vp)
    np.testing.assert_array_equal(a.target, b.target)
    lens = [len(i) for i in a.instr]
    assert lens == sorted(lens, reverse=True) and min(lens) >= 10 and max(lens) <= 79
    assert a.a_num.min() >= 2 and a.a_num.max() <= a.a_max
    live = a.target >= 0
    assert np.all(a.target[live] < a.a_num[live])
    assert np.all(live[0])                                    # step 0 is always supervised
    for bidx in range(16):                                    # -1 forever after the teacher's stop
        col = a.target[:, bidx]
        stop = np.where(col == 0)[0]
        if len(stop):
            assert np.all(col[stop[0] + 1:] == -1)
    t = synth.feature_table(3, 4)
    assert t.shape == (4, 36, 2048) and t.min() >= 0 and t.dtype == np.float32
    sb = synth.speaker_batch(seed=1, batch=5, n_viewpoints=10)
    assert np.array_equal(sb.act_is_stop.sum(0), np.ones(5))


def test_shard_rows_partitions_the_batch():
    from speaker_follower_amd import dp
    for n, w in ((100, 8), (100, 3), (7, 8), (64, 4)):
        seen = []
        for r in range(w):
            s = dp.shard_rows(n, r, w)
            seen += list(range(n))[s]
        assert seen == list(range(n))


def test_step_losses_global_normaliser_equals_unsharded_mean():
    """Sharded loss with the all-reduced (sum, count) table == the reference's batch mean."""
    from speaker_follower_amd import dp
    rng = np.random.default_rng(0)
    terms = rng.random((6, 10)).astype(np.float32)
    live = (rng.random((6, 10)) > 0.4).astype(np.float32)
    live[5] = 0                                               # a step with no live rows
    terms *= live
    full = torch.tensor(np.stack((terms.sum(1), live.sum(1)), 1))
    want = sum(terms[t].sum() / live[t].sum() for t in range(6) if live[t].sum() > 0)
    loss, gscale = dp.step_losses(full)
    np.testing.assert_allclose(float(loss), want, rtol=1e-6)
    assert float(gscale[5]) == 0.0
    a = torch.tensor(np.stack((terms[:, :4].sum(1), live[:, :4].sum(1)), 1))
    b = torch.tensor(np.stack((terms[:, 4:].sum(1), live[:, 4:].sum(1)), 1))
    loss2, _ = dp.step_losses(a + b)
    np.testing.assert_allclose(float(loss2), want, rtol=1e-6)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _spawn_world(worker, world, out):
    """mp.spawn of a gloo world on a free port; the port is found by bind-and-release, so another process can take it
    before rank 0 binds it: a rendezvous that fails for THAT reason is retried on a fresh port."""
    import torch.multiprocessing as mp
    for attempt in range(3):
        try:
            mp.spawn(worker, args=(world, _free_port(), out), nprocs=world, join=True)
            return
        except Exception as e:                                     # noqa: BLE001
            text = str(e).lower()
            if attempt == 2 or not any(s in text for s in ('address already in use', 'eaddrinuse', 'connection refused',
                                                           'connection reset', 'timed out')):
                raise
            out.clear()


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    from speaker_follower_amd import dp
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 4)
    frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
    fg = dp.FlatGrads(list(lin.parameters()) + [frozen])
    assert fg.attached() and fg.flat.numel() == 6 * 4 + 4
    # "backward": accumulate in place into the views, like the HIP kernels do
    lin.weight.grad += float(rank + 1)
    lin.bias.grad += 10.0 * (rank + 1)
    fg.allreduce()
    sc = torch.tensor([[1.0 + rank, 2.0], [0.0, 0.0]])
    dp.allreduce_step_counts(sc)
    other = [torch.nn.Parameter(torch.ones(5))]
    other[0].grad = torch.full((5,), float(rank))
    dp.allreduce_gradients(other)
    out[rank] = (lin.weight.grad.clone(), lin.bias.grad.clone(), sc, other[0].grad.clone(),
                 fg.attached())
    lin.zero_grad(set_to_none=True)
    try:
        fg.allreduce()
        out[rank] += (False,)
    except RuntimeError:
        out[rank] += (True,)
    dist.destroy_process_group()


def test_data_parallel_gradient_and_count_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    _spawn_world(_dp_worker, world, out)
    for r in range(world):
        w, b, sc, o, attached, raised = out[r]
        assert torch.all(w == 3.0) and torch.all(b == 30.0)        # 1+2, 10+20: SUM not mean
        assert sc.tolist() == [[3.0, 4.0], [0.0, 0.0]]
        assert torch.all(o == 1.0)
        assert attached and raised


def _bucket_worker(rank, world, port, out):
    import torch.distributed as dist
    from speaker_follower_amd import dp
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    dec_lstm, dec_rest, enc = torch.nn.Linear(8, 6), torch.nn.Linear(5, 3), torch.nn.Linear(4, 2)
    frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
    bk = dp.BucketedGrads([list(dec_lstm.parameters()), list(dec_rest.parameters()) + [frozen], list(enc.parameters())])
    one = dp.FlatGrads([torch.nn.Parameter(torch.zeros(bk.flat.numel()))])
    g = torch.Generator().manual_seed(100 + rank)
    vals = torch.randn(bk.flat.numel(), generator=g)
    bk.flat.copy_(vals)                       # "backward": the kernels accumulate into the views
    one.flat.copy_(vals)
    # production order of the backward: decoder LSTM, other decoder weights, encoder
    for b in range(bk.n_buckets):
        bk.launch(b)
    bk.wait()
    one.allreduce()
    reduced = bk.flat.clone()                 # (snapshot BEFORE the refusal stage below reduces bucket 0 a second time)
    again = None
    try:
        bk.launch(0)
        bk.wait()                             # buckets 1, 2 never launched: the optimizer must not run
    except RuntimeError as e:
        again = 'never launched' in str(e)
    # the refusal left nothing in flight: the buffer is quiescent and holds exactly one more reduction of bucket 0
    quiet = (bk._works == [] and bk.launched == [])
    lo, hi = bk.bounds[0]
    expect = reduced.clone()
    expect[lo:hi] *= world
    quiet = quiet and torch.equal(bk.flat, expect)
    for b in range(bk.n_buckets):             # ... and the object is usable again
        bk.launch(b)
    bk.wait()
    out[rank] = (reduced, one.flat.clone(), bk.bounds, bk.attached(), again,
                 dec_lstm.weight.grad.data_ptr() == bk.flat.data_ptr(), quiet)
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_equals_single_buffer_gloo_world2():
    """dp.BucketedGrads: buckets laid out in production order, each reduced by its own async all-reduce,
    give bit for bit the sum one all-reduce of the whole buffer gives; a backward that did not reach
    every bucket is refused at wait()."""
    import torch.multiprocessing as mp
    world = 2
    out = mp.Manager().dict()
    _spawn_world(_bucket_worker, world, out)
    for r in range(world):
        bucketed, single, bounds, attached, refused, first, quiet = out[r]
        assert torch.equal(bucketed, single)
        assert bounds == [(0, 54), (54, 72), (72, 82)]            # 8*6+6 | 5*3+3 | 4*2+2
        assert attached and refused and first and quiet
    assert torch.equal(out[0][0], out[1][0])


def _uneven_worker(rank, world, port, out):
    """SURVEY 8(e) end to end on the host: ONE global batch of 100 rows split with dp.shard_rows (uneven: 100 = 8 x 12 + 4),
    the per-step (CE sum, live count) table all-reduced, every rank's loss scaled by the GLOBAL count, gradients summed
    through the production-order buckets.  The result must be the gradient of the unsharded batch's loss
    (follower.py:278, 481: per-step mean over the non-ignored rows of the whole batch, summed over steps)."""
    import torch.distributed as dist
    import torch.nn.functional as Fn
    from speaker_follower_amd import dp
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    B, S, A, F = 100, 6, 5, 7
    g = torch.Generator().manual_seed(7)
    x = torch.randn(S, B, F, generator=g)
    target = torch.randint(0, A, (S, B), generator=g)
    ended = torch.rand(S, B, generator=g) < torch.linspace(0.0, 0.8, S)[:, None]
    ended[3, :] = True                                            # a step with NO live row anywhere: contributes 0
    ended[3 + 1:, :13] = False
    target = torch.where(ended, torch.full_like(target, -1), target)
    torch.manual_seed(3)
    m1, m2, m3 = torch.nn.Linear(F, 9), torch.nn.Linear(9, A), torch.nn.Linear(F, A)

    def terms(rows):
        h = torch.tanh(m1(x[:, rows]))
        logit = m2(h) + m3(x[:, rows])
        ce = Fn.cross_entropy(logit.reshape(-1, A), target[:, rows].reshape(-1), ignore_index=-1, reduction='none')
        return ce.reshape(S, -1).sum(1), (target[:, rows] >= 0).sum(1).to(torch.float32)

    params = [p for m in (m1, m2, m3) for p in m.parameters()]
    # the unsharded reference: follower.py's loss on all 100 rows
    s_all, c_all = terms(slice(0, B))
    live = c_all > 0
    full = (s_all[live] / c_all[live]).sum()
    want = torch.autograd.grad(full, params)
    # this rank's shard
    rows = dp.shard_rows(B, rank, world)
    bk = dp.BucketedGrads([list(m.parameters()) for m in (m2, m3, m1)])
    s_r, c_r = terms(rows)
    table = torch.stack([s_r.detach(), c_r], 1)
    dp.allreduce_step_counts(table)
    loss, gscale = dp.step_losses(table)
    mine = torch.autograd.grad((s_r * gscale).sum(), params)
    for p, gp in zip(params, mine):
        p.grad += gp                                               # in place, into the bucket buffer
    for b in range(bk.n_buckets):
        bk.launch(b)
    bk.wait()
    err = max(float((p.grad - w).abs().max()) / float(w.abs().max()) for p, w in zip(params, want))
    out[rank] = (rows.start, rows.stop, float(loss), float(full), err, table[:, 1].tolist(), c_all.tolist())
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [4, 8])
def test_uneven_row_shards_give_the_unsharded_gradient_gloo(world):
    import torch.multiprocessing as mp
    out = mp.Manager().dict()
    _spawn_world(_uneven_worker, world, out)
    covered = []
    for r in range(world):
        lo, hi, loss, full, err, counts, c_all = out[r]
        covered += list(range(lo, hi))
        assert abs(loss - full) <= 2e-6 * abs(full)               # every rank holds the GLOBAL loss
        assert err <= 2e-5                                         # (fp32 sums in a different order)
        assert counts == c_all and counts[3] == 0.0
    assert covered == list(range(100))                             # contiguous, disjoint, complete
    sizes = [out[r][1] - out[r][0] for r in range(world)]
    assert max(sizes) - min(sizes) <= (1 if 100 % world else 0)


def test_feature_store_reads_reference_tsv_format(tmp_path):
    """N4: the ResNet TSV of the reference (env.py:359-370: scanId, viewpointId, image_w, image_h,
    vfov, base64 fp32 36x2048) -> table + id index."""
    import base64
    from speaker_follower_amd.features import FeatureStore
    rng = np.random.default_rng(5)
    rows = {('scanA', 'vp%d' % i): rng.random((36, 2048), dtype=np.float32) for i in range(3)}
    path = tmp_path / 'feats.tsv'
    with open(path, 'wt') as f:
        for (scan, vp), feat in rows.items():
            f.write('\t'.join([scan, vp, '640', '480', '60', base64.b64encode(feat.tobytes()).decode()]) + '\n')
    store = FeatureStore.from_tsv(str(path), device='cpu')
    assert store.table.shape == (3, 36, 2048) and store.F == 2176
    for (scan, vp), feat in rows.items():
        np.testing.assert_array_equal(store.table[store.row(scan, vp)].numpy(), feat)
    assert store.loc_table.shape == (36, 36, 128)


def test_tsv_to_bin_round_trip(tmp_path):
    """N4: TSV -> flat .bin + id index -> FeatureStore, bit-exact with the TSV reader."""
    import base64
    from speaker_follower_amd.features import FeatureStore, tsv_to_bin
    rng = np.random.default_rng(6)
    feats = [rng.random((36, 2048), dtype=np.float32) for _ in range(5)]
    tsv = tmp_path / 'f.tsv'
    with open(tsv, 'wt') as f:
        for i, feat in enumerate(feats):
            f.write('\t'.join(['scan%d' % (i % 2), 'v%d' % i, '640', '480', '60',
                               base64.b64encode(feat.tobytes()).decode()]) + '\n')
    assert tsv_to_bin(str(tsv), str(tmp_path / 'f.bin')) == 5
    assert os.path.getsize(tmp_path / 'f.bin') == 5 * 36 * 2048 * 4
    a = FeatureStore.from_tsv(str(tsv), device='cpu')
    b = FeatureStore.from_bin(str(tmp_path / 'f.bin'), device='cpu', chunk_rows=2)
    assert a.index == b.index
    assert torch.equal(a.table, b.table)
    bad = tmp_path / 'bad.tsv'
    with open(bad, 'wt') as f:
        f.write('\t'.join(['s', 'v', '640', '480', '60', base64.b64encode(b'1234').decode()]) + '\n')
    with pytest.raises(ValueError):
        tsv_to_bin(str(bad), str(tmp_path / 'bad.bin'))


def test_checkpoint_files_and_result_json_follow_the_reference_layout(tmp_path):
    """N4: `<prefix>_enc` / `<prefix>_dec` torch.save(state_dict) pairs (follower.py:1022-1035) and
    the result file {instr_id: {instr_id, trajectory}} (follower.py:117-125) that eval.py reads."""
    import json
    from speaker_follower_amd import model, agents, synth
    d = synth.SMALL
    enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5)
    dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    agent = agents.Seq2SeqAgent(None, str(tmp_path / 'res.json'), enc, dec)
    agent.save(str(tmp_path / 'snap'))
    assert os.path.exists(tmp_path / 'snap_enc') and os.path.exists(tmp_path / 'snap_dec')
    sd = torch.load(tmp_path / 'snap_dec')
    assert set(sd) == set(dec.state_dict()) and 'lstm.weight_ih' in sd
    enc2 = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5)
    dec2 = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
    agent2 = agents.Seq2SeqAgent(None, '', enc2, dec2)
    agent2.load(str(tmp_path / 'snap'), map_location='cpu')
    for (k, a), (_, b) in zip(dec.state_dict().items(), dec2.state_dict().items()):
        assert torch.equal(a, b), k
    agent.results = {'7_0': dict(instr_id='7_0', trajectory=[('vp', 0.0, 0.0)], score=-1.0,
                                 observations=['not serialisable' for _ in range(2)])}
    agent.write_results()
    out = json.load(open(tmp_path / 'res.json'))
    assert out == {'7_0': {'instr_id': '7_0', 'trajectory': [['vp', 0.0, 0.0]]}}


def test_packed_speaker_batch_equals_the_field_by_field_upload():
    """speaker.pack_speaker_batch / batch_from_packed (one buffer, one H2D copy per minibatch: the configs[2] sweep)
    hold exactly what DeviceSpeakerBatch.from_synth uploads field by field -- ragged paths, instructions longer than
    the word budget included."""
    import torch
    from speaker_follower_amd import synth, speaker
    sb = synth.speaker_batch(seed=11, batch=9, n_viewpoints=50, min_path=2, max_path=6, min_len=3, max_len=40)
    Lmax = 24                                       # some instructions are longer than this: truncation path
    want = speaker.DeviceSpeakerBatch.from_synth(sb, device='cpu', max_length=Lmax)
    Tp = int(sb.path_len.max())
    buf = torch.zeros(speaker.packed_layout(9, Tp, Lmax)['bytes'], dtype=torch.uint8)
    B, Tp2 = speaker.pack_speaker_batch(sb, buf.numpy(), Lmax)
    assert (B, Tp2) == (9, Tp)
    got = speaker.batch_from_packed(buf, B, Tp, Lmax)
    for f in ('instr_seq', 'path_mask', 'view', 'act', 'act_view', 'act_sincos'):
        a, b = getattr(got, f), getattr(want, f)
        assert a.dtype == b.dtype and torch.equal(a, b[:Tp] if f not in ('instr_seq', 'path_mask') else b[:, :a.shape[1]]), f
    assert torch.equal(got.vp, want.vp[:Tp])
