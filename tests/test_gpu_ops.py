"""GPU parity: every C-ABI operator against the oracle on the same seeded inputs.
Tolerances: 1e-4 on logits / activations (BASELINE.json north_star), bit-exact argmax."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import np_env, np_model, rng as orng, torch_ref          # noqa: E402
from speaker_follower_amd import synth                                # noqa: E402

TOL = dict(rtol=1e-4, atol=1e-4)


def grad_close(got, ref, name=''):
    """Weight gradients are sums over the batch: compare at 2e-5 of the tensor's own scale."""
    got = got.detach().cpu().numpy() if hasattr(got, 'detach') else np.asarray(got)
    ref = ref.detach().cpu().numpy() if hasattr(ref, 'detach') else np.asarray(ref)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5 * max(1.0, float(np.abs(ref).max())),
                               err_msg=name)


@pytest.fixture(scope='module')
def sf():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    from speaker_follower_amd import _lib, ops, model, features, follower
    import types
    return types.SimpleNamespace(lib=_lib, ops=ops, model=model, features=features,
                                 follower=follower)


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


# ------------------------------------------------------------------------------------ GEMMs
@pytest.mark.parametrize('M,N,K', [(100, 2048, 4864), (8, 256, 2176), (128, 2048, 4352), (33, 64, 2048),
                                   (100, 2048, 512), (8, 256, 512), (3, 512, 1024), (100, 991, 512),
                                   (7, 2048, 300), (128, 64, 16), (250, 2048, 300), (1, 16, 4),
                                   # deep reductions: the LDS-tiled weight-gradient kernel (M >= 4096),
                                   # with a ragged last column tile
                                   (4500, 512, 256), (4100, 256, 304)])
@pytest.mark.parametrize('act', [0, 1])
def test_linear_fwd_bwd(sf, M, N, K, act):
    rng = np.random.default_rng(M * 7 + N + K)
    x, w, b = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5), rnd(rng, N)
    y = sf.ops.linear_fwd(dev(x), dev(w), dev(b), act)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    if act:
        ref = np.tanh(ref)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-5, atol=2e-5)
    if N % 4 or K % 4:
        return
    dy = rnd(rng, M, N)
    dw = torch.zeros(N, K, device='cuda')
    db = torch.zeros(N, device='cuda')
    dx = sf.ops.linear_bwd(dev(x), dev(w), y, dev(dy), act, True, dw, db)
    dpre = dy.astype(np.float64) * ((1 - ref ** 2) if act else 1.0)
    np.testing.assert_allclose(dx.cpu().numpy(), dpre @ w, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dw.cpu().numpy(), dpre.T @ x, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), dpre.sum(0), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('M,N,K', [(2000, 256, 2176), (300, 128, 128), (333, 256, 128), (8000, 2048, 512),
                                   (2000, 2048, 4352), (257, 384, 256)])
def test_weight_gradient_on_the_bf16_matrix_cores(sf, M, N, K):
    """dW [N,K] += dY^T X through gemm_tn_split_kernel (bf16x6 split product, operands transposed on their way into
    LDS; split-M slabs for small outputs; ragged last stage) against float64: accumulated INTO a non-zero dW, asymmetric
    operands (a swapped fragment mapping or a wrong swizzle cannot pass), accuracy better than the fp32-MFMA kernel's."""
    rng = np.random.default_rng(M + N + K)
    x = rnd(rng, M, K) + (np.arange(K, dtype=np.float32) % 7)[None, :] * 0.1
    dy = rnd(rng, M, N) * (1.0 + (np.arange(N, dtype=np.float32) % 5)[None, :])
    w = rnd(rng, N, K, scale=K ** -0.5)
    dw0 = rnd(rng, N, K)
    ref = dw0.astype(np.float64) + dy.astype(np.float64).T @ x.astype(np.float64)
    scale = float(np.abs(ref).max())
    errs = {}
    for f32_path in (0, 1):
        sf.lib.lib.sf_debug_gate_product_f32(f32_path)
        sf.lib.lib.sf_debug_tn_split_min_rows(256)          # (the dispatch uses the split kernel from 4 096 rows on)
        try:
            dw = dev(dw0.copy())
            sf.ops.linear_bwd(dev(x), dev(w), None, dev(dy), 0, False, dw, None)
            torch.cuda.synchronize()
        finally:
            sf.lib.lib.sf_debug_gate_product_f32(0)
            sf.lib.lib.sf_debug_tn_split_min_rows(-1)
        errs[f32_path] = float(np.abs(dw.cpu().numpy() - ref).max()) / scale
    print('[wgrad %dx%dx%d] max error / scale: bf16x6 %.2e, fp32 MFMA %.2e' % (M, N, K, errs[0], errs[1]))
    assert errs[0] <= 2e-6 and errs[0] <= 1.5 * errs[1] + 1e-7


@pytest.mark.parametrize('M,N,K,act', [(8000, 512, 512, 0), (8000, 512, 1024, 1), (8000, 991, 512, 0), (2560, 512, 512, 0),
                                       (515, 130, 96, 1), (4100, 64, 64, 0), (1000, 2048, 992, 0),
                                       (991, 2048, 300, 0), (4100, 256, 304, 0)])       # partial last stage (K % 32 != 0)
def test_many_row_product_on_the_bf16_matrix_cores(sf, M, N, K, act):
    """y = act(x W^T + b) for M >= 512 through gemm_nt_big_kernel (128 x 128 tiles, both operands staged through LDS as
    three bf16 planes) against float64: asymmetric, badly scaled operands (a swapped fragment mapping, a wrong chunk
    order or a dropped plane cannot pass), ragged last row / column tiles, and not worse than the register-streaming
    kernel of rounds 1-4."""
    rng = np.random.default_rng(M + N + K)
    x = (rnd(rng, M, K) * (1.0 + 3.0 * (np.arange(K) % 7 == 0))[None, :] + 0.25).astype(np.float32)
    w = (rnd(rng, N, K, scale=K ** -0.5) * (1.0 + (np.arange(N) % 5)[:, None])).astype(np.float32)
    b = rnd(rng, N)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    pre_scale = float(np.abs(ref).max())
    if act:
        ref = np.tanh(ref)
    errs = {}
    for big in (1, 0):
        sf.lib.lib.sf_debug_many_row_product(big)
        try:
            y = sf.ops.linear_fwd(dev(x), dev(w), dev(b), act)
            torch.cuda.synchronize()
        finally:
            sf.lib.lib.sf_debug_many_row_product(1)
        errs[big] = float(np.abs(y.cpu().numpy() - ref).max()) / float(np.abs(ref).max())
    print('[many-row product %dx%dx%d] max error / scale: LDS-tiled bf16x6 %.2e, register-streaming fp32 MFMA %.2e'
          % (M, N, K, errs[1], errs[0]))
    # (with tanh the error of the pre-activation -- up to |x W^T| ~ 30 here -- passes through a slope <= 1)
    assert errs[1] <= 1e-6 * (max(1.0, pre_scale) if act else 1.0) and errs[1] <= 1.5 * errs[0] + 1e-7


@pytest.mark.parametrize('M,N,K', [(2000, 256, 2176), (2000, 512, 1024), (2000, 256, 512), (1024, 100, 300), (8000, 992, 512),
                                   (2004, 64, 64)])
def test_small_weight_gradient_through_the_many_row_kernel(sf, M, N, K):
    """dW [N,K] += dY^T X for a small weight over >= 1 024 stacked rows (the decoder's eight small weight gradients over
    S*B = 2 000 rows): two tiled transposes + gemm_nt_big_kernel with a K-split, against float64 -- accumulated INTO a
    non-zero dW, asymmetric operands, ragged tiles, a reduction that is not a multiple of the stage depth; at least as
    accurate as the kernels of rounds 1-4."""
    rng = np.random.default_rng(M + N + K)
    x = (rnd(rng, M, K) + (np.arange(K) % 7)[None, :] * 0.1).astype(np.float32)
    dy = (rnd(rng, M, N) * (1.0 + (np.arange(N) % 5)[None, :])).astype(np.float32)
    w = rnd(rng, N, K, scale=K ** -0.5)
    dw0 = rnd(rng, N, K)
    ref = dw0.astype(np.float64) + dy.astype(np.float64).T @ x.astype(np.float64)
    scale = float(np.abs(ref).max())
    errs = {}
    for big in (1, 0):
        sf.lib.lib.sf_debug_many_row_product(big)
        try:
            dw = dev(dw0.copy())
            sf.ops.linear_bwd(dev(x), dev(w), None, dev(dy), 0, False, dw, None)
            torch.cuda.synchronize()
        finally:
            sf.lib.lib.sf_debug_many_row_product(1)
        errs[big] = float(np.abs(dw.cpu().numpy() - ref).max()) / scale
    print('[small wgrad %dx%dx%d] max error / scale: transposes + bf16x6 tiles %.2e, rounds 1-4 kernels %.2e' % (M, N, K, errs[1], errs[0]))
    assert errs[1] <= 2e-6 and errs[1] <= 1.5 * errs[0] + 1e-7


def test_gemm_is_transpose_safe(sf):
    """Asymmetric operands: catches a row/column swap in the MFMA fragment mapping."""
    M, N, K = 48, 80, 32
    x = np.arange(M * K, dtype=np.float32).reshape(M, K) / 97.0
    w = (np.arange(N * K, dtype=np.float32).reshape(N, K) % 13) / 5.0 - 1.0
    y = sf.ops.linear_fwd(dev(x), dev(w)).cpu().numpy()
    np.testing.assert_allclose(y, x.astype(np.float64) @ w.astype(np.float64).T, rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------------ LSTMCell
@pytest.mark.parametrize('B,I,H', [(8, 4352, 512), (100, 4352, 512), (5, 300, 512), (3, 48, 16),
                                   # B > 16: the 32-row x 8-unit recurrent step, with an input segment
                                   (100, 300, 512), (33, 300, 512), (17, 48, 16)])
def test_lstm_cell(sf, B, I, H):
    rng = np.random.default_rng(B + I)
    k = H ** -0.5
    w = [rnd(rng, 4 * H, I, scale=k), rnd(rng, 4 * H, H, scale=k), rnd(rng, 4 * H, scale=k),
         rnd(rng, 4 * H, scale=k)]
    x, h, c = rnd(rng, B, I), rnd(rng, B, H), rnd(rng, B, H)
    wd = [dev(a) for a in w]
    h1, c1, gates = sf.ops.lstm_cell_fwd(wd, dev(x), dev(h), dev(c))
    rh, rc = np_model.lstm_cell(x, h, c, *w)
    np.testing.assert_allclose(h1.cpu().numpy(), rh, **TOL)
    np.testing.assert_allclose(c1.cpu().numpy(), rc, **TOL)
    # backward against torch autograd
    tw = [torch.tensor(a, requires_grad=True) for a in w]
    tx, th, tc = (torch.tensor(a, requires_grad=True) for a in (x, h, c))
    th1, tc1 = torch_ref.lstm_cell(tx, th, tc, *tw)
    gh, gc = rnd(rng, B, H), rnd(rng, B, H)
    ((th1 * torch.tensor(gh)).sum() + (tc1 * torch.tensor(gc)).sum()).backward()
    g = [torch.zeros_like(a) for a in wd]
    dx, dh0, dc0 = sf.ops.lstm_cell_bwd(wd, g, dev(x), dev(h), dev(c), c1, gates, dev(gh), dev(gc))
    np.testing.assert_allclose(dx.cpu().numpy(), tx.grad.numpy(), **TOL)
    np.testing.assert_allclose(dh0.cpu().numpy(), th.grad.numpy(), **TOL)
    np.testing.assert_allclose(dc0.cpu().numpy(), tc.grad.numpy(), **TOL)
    for got, ref in zip(g, tw):
        grad_close(got, ref.grad)


# ------------------------------------------------------------------------------------ attentions
def _small_modules_golden(golden, prefix):
    g = golden('g1_modules_small')
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def test_visual_attention_module_small_golden(sf, golden):
    """VisualSoftDotAttention module (dense API) on the reference's own small-dim vectors."""
    g = _small_modules_golden(golden, 'vsda/')
    H, F, D = g['h'].shape[1], g['X'].shape[2], g['linear_in_h.weight'].shape[0]
    m = sf.model.VisualSoftDotAttention(H, F, dot_dim=D).cuda()
    m.load_state_dict({k: torch.tensor(g[k]) for k in
                       ('linear_in_h.weight', 'linear_in_h.bias', 'linear_in_v.weight', 'linear_in_v.bias')})
    h = dev(g['h']).requires_grad_(True)
    out, alpha = m(h, dev(g['X']))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g['out'], **TOL)
    np.testing.assert_allclose(alpha.cpu().numpy(), g['alpha'], **TOL)
    (out * dev(g['go'])).sum().backward()
    np.testing.assert_allclose(h.grad.cpu().numpy(), g['dh'], **TOL)
    np.testing.assert_allclose(m.linear_in_h.weight.grad.cpu().numpy(), g['d_linear_in_h.weight'], **TOL)
    np.testing.assert_allclose(m.linear_in_h.bias.grad.cpu().numpy(), g['d_linear_in_h.bias'], **TOL)
    np.testing.assert_allclose(m.linear_in_v.weight.grad.cpu().numpy(), g['d_linear_in_v.weight'], **TOL)


def test_soft_dot_attention_module_small_golden(sf, golden):
    g = _small_modules_golden(golden, 'sda/')
    H = g['h'].shape[1]
    m = sf.model.SoftDotAttention(H).cuda()
    m.load_state_dict({k: torch.tensor(g[k]) for k in ('linear_in.weight', 'linear_out.weight')})
    h = dev(g['h']).requires_grad_(True)
    ctx = dev(g['ctx']).requires_grad_(True)
    ht, alpha = m(h, ctx, dev(g['mask']))
    np.testing.assert_allclose(ht.detach().cpu().numpy(), g['h_tilde'], **TOL)
    np.testing.assert_allclose(alpha.cpu().numpy(), g['alpha'], **TOL)
    (ht * dev(g['go'])).sum().backward()
    np.testing.assert_allclose(h.grad.cpu().numpy(), g['dh'], **TOL)
    np.testing.assert_allclose(ctx.grad.cpu().numpy(), g['dctx'], **TOL)
    np.testing.assert_allclose(m.linear_in.weight.grad.cpu().numpy(), g['d_linear_in.weight'], **TOL)
    np.testing.assert_allclose(m.linear_out.weight.grad.cpu().numpy(), g['d_linear_out.weight'], **TOL)


def test_eltwise_prod_scoring_module_small_golden(sf, golden):
    g = _small_modules_golden(golden, 'eps/')
    H, F, D = g['h'].shape[1], g['U'].shape[2], g['linear_in_h.weight'].shape[0]
    m = sf.model.EltwiseProdScoring(H, F, dot_dim=D).cuda()
    names = ('linear_in_h.weight', 'linear_in_h.bias', 'linear_in_a.weight', 'linear_in_a.bias',
             'linear_out.weight', 'linear_out.bias')
    m.load_state_dict({k: torch.tensor(g[k]) for k in names})
    h = dev(g['h']).requires_grad_(True)
    logit = m(h, dev(g['U']))
    np.testing.assert_allclose(logit.detach().cpu().numpy(), g['logit'], **TOL)
    (logit * dev(g['go'])).sum().backward()
    np.testing.assert_allclose(h.grad.cpu().numpy(), g['dh'], **TOL)
    for n in names:
        mod, attr = n.split('.')
        got = getattr(getattr(m, mod), attr).grad.cpu().numpy()
        np.testing.assert_allclose(got, g['d_' + n], rtol=1e-4, atol=1e-4, err_msg=n)


@pytest.mark.parametrize('B', [5, 100])
def test_visual_attention_full_dims_dense_and_indexed(sf, B):
    rng = np.random.default_rng(B)
    d = synth.FULL
    H, F, D, V = d.hidden, d.feat, d.dot, d.views
    table = synth.feature_table(3, 40)
    loc = np_env.static_loc_embeddings()
    vp = rng.integers(0, 40, B).astype(np.int32)
    view = rng.integers(0, 36, B).astype(np.int32)
    X = np.stack([np_env.panorama_feature(table[vp[b]], view[b], loc) for b in range(B)])
    w = [rnd(rng, D, H, scale=H ** -0.5), rnd(rng, D, scale=0.1), rnd(rng, D, F, scale=F ** -0.5),
         rnd(rng, D, scale=0.1)]
    h = rnd(rng, B, H)
    ref_out, ref_alpha = np_model.visual_soft_dot_attention(h, X, *w)
    wd = [dev(a) for a in w]
    store = sf.features.FeatureStore(table)
    np.testing.assert_array_equal(store.loc_table.cpu().numpy(), loc)
    vp_d, view_d = dev(vp), dev(view)          # index tensors must outlive the enqueued kernels
    Xg = store.gather_panorama(vp_d, view_d)
    np.testing.assert_array_equal(Xg.cpu().numpy(), X)                        # a11 gather: bit-exact
    for pano in (sf.ops.pano_dense(Xg), store.pano(vp_d, view_d)):
        out, alpha, t_v, q = sf.ops.visual_attention_fwd(wd, pano, B, V, F, dev(h))
        np.testing.assert_allclose(out.cpu().numpy(), ref_out, **TOL)
        np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, **TOL)
    # backward vs autograd
    tw = [torch.tensor(a, requires_grad=True) for a in w]
    th = torch.tensor(h, requires_grad=True)
    tout, _ = torch_ref.visual_soft_dot_attention(th, torch.tensor(X), *tw)
    go = rnd(rng, B, F)
    (tout * torch.tensor(go)).sum().backward()
    g = [torch.zeros_like(a) for a in wd]
    dh = sf.ops.visual_attention_bwd(wd, g, store.pano(vp_d, view_d), B, dev(h), alpha, t_v,
                                     dev(go))
    np.testing.assert_allclose(dh.cpu().numpy(), th.grad.numpy(), **TOL)
    for i in range(3):
        grad_close(g[i], tw[i].grad, 'visual w%d' % i)
    assert float(g[3].abs().max()) == 0.0          # b_v: exactly zero by construction


@pytest.mark.parametrize('B,L', [(6, 80), (100, 37), (4, 7), (3, 1), (5, 128)])
def test_soft_dot_attention_full_dims(sf, B, L):
    rng = np.random.default_rng(B * 100 + L)
    H = 512
    w_in, w_out = rnd(rng, H, H, scale=H ** -0.5), rnd(rng, H, 2 * H, scale=(2 * H) ** -0.5)
    h, ctx = rnd(rng, B, H), rnd(rng, B, L, H)
    lens = rng.integers(1, L + 1, B)
    lens[0] = L
    mask = np.arange(L)[None, :] >= lens[:, None]
    ref_ht, ref_alpha = np_model.soft_dot_attention(h, ctx, mask, w_in, w_out)
    wd = (dev(w_in), dev(w_out))
    ht, alpha, cat2, t_text = sf.ops.soft_dot_attention_fwd(wd, dev(h), dev(ctx), dev(mask))
    np.testing.assert_allclose(ht.cpu().numpy(), ref_ht, **TOL)
    np.testing.assert_allclose(alpha.cpu().numpy(), ref_alpha, **TOL)
    assert float(alpha.cpu().numpy()[mask].sum()) == 0.0
    tw = [torch.tensor(w_in, requires_grad=True), torch.tensor(w_out, requires_grad=True)]
    th, tctx = torch.tensor(h, requires_grad=True), torch.tensor(ctx, requires_grad=True)
    tht, _ = torch_ref.soft_dot_attention(th, tctx, torch.tensor(mask), *tw)
    go = rnd(rng, B, H)
    (tht * torch.tensor(go)).sum().backward()
    g = (torch.zeros_like(wd[0]), torch.zeros_like(wd[1]))
    dh, dctx = sf.ops.soft_dot_attention_bwd(wd, g, dev(ctx), alpha, cat2, t_text, ht, dev(go))
    np.testing.assert_allclose(dh.cpu().numpy(), th.grad.numpy(), **TOL)
    np.testing.assert_allclose(dctx.cpu().numpy(), tctx.grad.numpy(), **TOL)
    grad_close(g[0], tw[0].grad, 'w_in')
    grad_close(g[1], tw[1].grad, 'w_out')


@pytest.mark.parametrize('B,A', [(7, 14), (100, 9), (3, 2), (4, 16)])
def test_scoring_full_dims_dense_and_indexed(sf, B, A):
    rng = np.random.default_rng(B + 31 * A)
    d = synth.FULL
    H, F, D = d.hidden, d.feat, d.dot
    fb = synth.follower_batch(seed=B, batch=B, steps=1, n_viewpoints=30, a_max=A)
    table = synth.feature_table(5, 30)
    loc = np_env.static_loc_embeddings()
    X, all_u, is_valid = np_env.dense_follower_step(table, loc, fb, 0)
    pad = np.zeros((B, A, F), np.float32)
    pad[:, :all_u.shape[1]] = all_u
    w = [rnd(rng, D, H, scale=H ** -0.5), rnd(rng, D, scale=0.1), rnd(rng, D, F, scale=F ** -0.5),
         rnd(rng, D, scale=0.1), rnd(rng, 1, D, scale=D ** -0.5), rnd(rng, 1, scale=0.1)]
    h = rnd(rng, B, H)
    ref = np_model.eltwise_prod_scoring(h, pad, *w)
    wd = [dev(a) for a in w]
    store = sf.features.FeatureStore(table)
    idx = [dev(fb.vp[0]), dev(fb.cand_view[0]),
           dev(sf.features.cand_sincos(fb.cand_heading[0], fb.cand_elevation[0])), dev(fb.a_num[0])]
    Ug, validg = store.gather_candidates(*idx)
    np.testing.assert_array_equal(Ug.cpu().numpy(), pad)                      # a11: bit-exact
    np.testing.assert_array_equal(validg.cpu().numpy()[:, :is_valid.shape[1]], is_valid)
    for cnd in (sf.ops.cands_dense(Ug), store.cands(*idx, A)):
        logit, t_a, wt, r = sf.ops.eltwise_prod_scoring_fwd(wd, cnd, B, A, F, dev(h))
        np.testing.assert_allclose(logit.cpu().numpy(), ref, **TOL)
    tw = [torch.tensor(a, requires_grad=True) for a in w]
    th = torch.tensor(h, requires_grad=True)
    go = rnd(rng, B, A)
    (torch_ref.eltwise_prod_scoring(th, torch.tensor(pad), *tw) * torch.tensor(go)).sum().backward()
    g = [torch.zeros_like(a) for a in wd]
    dh = sf.ops.eltwise_prod_scoring_bwd(wd, g, store.cands(*idx, A), B, dev(h), t_a, wt, dev(go))
    np.testing.assert_allclose(dh.cpu().numpy(), th.grad.numpy(), **TOL)
    for got, refp in zip(g, tw):
        grad_close(got, refp.grad)


def test_dropout_mask_matches_oracle_mirror(sf):
    import ctypes as C
    from speaker_follower_amd.runtime import ptr, stream, dropout_arg
    B, N = 5, 300
    x = torch.ones(B, N, device='cuda')
    out = torch.empty_like(x)
    sf.lib.call('sf_dropout_copy', ptr(x), N, B, N, ptr(out), N, dropout_arg(0.5, 1234, 17), 9, 40,
                stream())
    ref = orng.dropout_mask(1234, 9, np.arange(17, 17 + B), 40 + N, 0.5)[:, 40:]
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    assert 0.4 < (ref == 0).mean() < 0.6


def test_fill_regions_sets_every_buffer_and_nothing_else(sf):
    """sf_fill_regions: the initial conditions of a pass (zero states, <BOS> words, cleared flags) in one launch."""
    from speaker_follower_amd.runtime import fill_regions
    big = torch.full((3, 100, 512), 7.0, device='cuda')
    words = torch.full((5, 101), -3, dtype=torch.int64, device='cuda')
    ended = torch.full((103,), 9, dtype=torch.uint8, device='cuda')
    idx = torch.full((2, 37), 11, dtype=torch.int32, device='cuda')
    long_one = torch.full((300001,), 1.0, device='cuda')                   # more elements than one pass of the grid
    fill_regions((big[1], -0.5), (words[0], 1), (ended[:101], 0), (idx[1], -2), (long_one, 2.25), (big[2, :0], 3.0))
    torch.cuda.synchronize()
    assert (big[0] == 7).all() and (big[1] == -0.5).all() and (big[2] == 7).all()
    assert (words[0] == 1).all() and (words[1:] == -3).all()
    assert (ended[:101] == 0).all() and (ended[101:] == 9).all()
    assert (idx[0] == 11).all() and (idx[1] == -2).all() and (long_one == 2.25).all()
    with pytest.raises(RuntimeError):
        fill_regions(*[(ended, 0)] * 9)                                    # at most SF_FILL_MAX_REGIONS
    fill_regions((words[4], 2 ** 40 + 5))                                  # all 8 bytes of a 64-bit value
    assert (words[4] == 2 ** 40 + 5).all()


def test_move_rows_is_the_separate_gathers_and_scatters(sf):
    """sf_move_rows against sf_gather_rows / sf_scatter_rows semantics: several moves in one launch."""
    import ctypes as C
    from speaker_follower_amd.runtime import stream
    rng = np.random.default_rng(5)
    n, H, T = 77, 512, 80
    hpool, cpool = dev(rnd(rng, 300, H)), dev(rnd(rng, 300, H))
    idx = rng.integers(-1, 300, n).astype(np.int32)
    idx_d = dev(idx)
    h0, c0 = torch.full((n, H), 9.0, device='cuda'), torch.full((n, H + 4), 9.0, device='cuda')
    L = sf.lib
    gat = (L.RowMove * 2)(L.RowMove(hpool.data_ptr(), h0.data_ptr(), idx_d.data_ptr(), H, H, H, 0),
                          L.RowMove(cpool.data_ptr(), c0.data_ptr(), idx_d.data_ptr(), H, H + 4, H, 0))
    L.call('sf_move_rows', gat, 2, n, stream())
    ref_h = np.where(idx[:, None] >= 0, hpool.cpu().numpy()[np.maximum(idx, 0)], 0.0)
    ref_c = np.where(idx[:, None] >= 0, cpool.cpu().numpy()[np.maximum(idx, 0)], 0.0)
    np.testing.assert_array_equal(h0.cpu().numpy(), ref_h)
    np.testing.assert_array_equal(c0[:, :H].cpu().numpy(), ref_c)
    assert (c0[:, H:] == 9).all()                                          # the padding columns are not touched
    # scatters: three widths, rows with idx < 0 skipped
    dst = np.where(rng.random(n) < 0.2, -1, rng.permutation(400)[:n]).astype(np.int32)
    dst_d = dev(dst)
    srcs = [dev(rnd(rng, n, w)) for w in (H, H, T)]
    pools = [torch.full((400, w), -1.0, device='cuda') for w in (H, H, T)]
    sca = (L.RowMove * 3)(*(L.RowMove(a.data_ptr(), b.data_ptr(), dst_d.data_ptr(), w, w, w, 1)
                            for a, b, w in zip(srcs, pools, (H, H, T))))
    L.call('sf_move_rows', sca, 3, n, stream())
    for a, b in zip(srcs, pools):
        ref = np.full(b.shape, -1.0, np.float32)
        ref[dst[dst >= 0]] = a.cpu().numpy()[dst >= 0]
        np.testing.assert_array_equal(b.cpu().numpy(), ref)
    with pytest.raises(RuntimeError):
        L.call('sf_move_rows', sca, 5, n, stream())                        # at most SF_ROW_MOVES_MAX


def test_linear_slabs_sum_to_the_product():
    """sf_linear_slabs_fwd: the gate product as split-K slabs (what bench.py's roofline object times)."""
    import ctypes as C
    from speaker_follower_amd._lib import call
    from speaker_follower_amd.runtime import ptr, ws_args, workspace
    g = torch.Generator().manual_seed(4)
    M, N, K1, K2 = 100, 2048, 4352, 512
    x, h = torch.randn(M, K1, generator=g).cuda(), torch.randn(M, K2, generator=g).cuda()
    w, u = (torch.randn(N, K1, generator=g) * 0.02).cuda(), (torch.randn(N, K2, generator=g) * 0.02).cuda()
    ks = C.c_int(0)
    call('sf_linear_slabs_fwd', ptr(x), K1, ptr(w), K1, ptr(h), K2, ptr(u), K2, M, N, C.byref(ks),
         *ws_args(x.device))
    torch.cuda.synchronize()
    assert ks.value >= 1
    slabs = workspace(x.device)[:ks.value * M * N * 4].view(torch.float32).view(ks.value, M, N)
    ref = x.double() @ w.double().T + h.double() @ u.double().T
    torch.testing.assert_close(slabs.sum(0).double(), ref, rtol=1e-4, atol=1e-4)


def test_many_row_gate_product_takes_k_splits_where_tiles_waste_a_round():
    """csrc/sf_gemm.hip linear_nt: the beam step's gate product (2 560 states: 20 x 16 = 320 tiles of 128 x 128 on 256 CUs)
    is issued as three K splits whose slabs the consumer adds up; the slabs' sum equals the unsplit product of the same
    kernel to rounding of the summation order, and both stay in the fp32-roundoff class against float64."""
    import ctypes as C
    from speaker_follower_amd._lib import call, lib
    from speaker_follower_amd.runtime import ptr, ws_args, workspace
    g = torch.Generator().manual_seed(9)
    M, N, K1, K2 = 2560, 2048, 4352, 512
    x = torch.relu(torch.randn(M, K1, generator=g) * 0.5 + 0.4).cuda()
    h = torch.tanh(torch.randn(M, K2, generator=g)).cuda()
    w, u = (torch.randn(N, K1, generator=g) * 0.03).cuda(), (torch.randn(N, K2, generator=g) * 0.05).cuda()
    ref = x.double() @ w.double().T + h.double() @ u.double().T
    mag = x.double().abs() @ w.double().abs().T + h.double().abs() @ u.double().abs().T
    got = {}
    for mode in (1, 3):
        lib.sf_debug_many_row_product(mode)
        try:
            ks = C.c_int(0)
            call('sf_linear_slabs_fwd', ptr(x), K1, ptr(w), K1, ptr(h), K2, ptr(u), K2, M, N, C.byref(ks),
                 *ws_args(x.device))
            torch.cuda.synchronize()
        finally:
            lib.sf_debug_many_row_product(1)
        slabs = workspace(x.device)[:ks.value * M * N * 4].view(torch.float32).view(ks.value, M, N)
        got[mode] = (ks.value, slabs.sum(0).double())
    assert got[1][0] == 3 and got[3][0] == 1, (got[1][0], got[3][0])
    for mode in (1, 3):
        rel = float(((got[mode][1] - ref).abs() / mag).max())
        print('[many-row gate product, %d K splits] max error / sum|a b| = %.2e' % (got[mode][0], rel))
        assert rel < 2.5e-7
    assert float(((got[1][1] - got[3][1]).abs() / mag).max()) < 2.5e-7


@pytest.mark.parametrize('M', [100, 1, 16, 37, 64, 113, 128])
def test_gate_product_on_the_bf16_matrix_cores_keeps_fp32_accuracy(M):
    """csrc/sf_gemm.hip: gemm_nt_split_kernel (three-way error-free bf16 splitting, six bf16 MFMAs per product,
    fp32 accumulate) against float64 and against the fp32-MFMA kernel it replaces (sf_debug_gate_product_f32): the
    error stays in the fp32-roundoff class -- bounded by 2.5e-7 sum|a b| -- and is no larger than the fp32 kernel's."""
    import ctypes as C
    from speaker_follower_amd._lib import call, lib, kernel_profile
    from speaker_follower_amd.runtime import ptr, ws_args, workspace
    g = torch.Generator().manual_seed(M)
    N, K1, K2 = 2048, 4352, 512
    # [u | feature] post-ReLU non-negative with dropout (x2 or 0), h in (-1, 1): the decoder's LSTM input
    x = (torch.relu(torch.randn(M, K1, generator=g) * 0.5 + 0.4) * 2 * (torch.rand(M, K1, generator=g) < 0.5)).cuda()
    h = torch.tanh(torch.randn(M, K2, generator=g)).cuda()
    w, u = (torch.randn(N, K1, generator=g) * 0.03).cuda(), (torch.randn(N, K2, generator=g) * 0.05).cuda()
    ref = x.double() @ w.double().T + h.double() @ u.double().T
    mag = x.double().abs() @ w.double().abs().T + h.double().abs() @ u.double().abs().T
    out = {}
    for f32 in (1, 0):
        lib.sf_debug_gate_product_f32(f32)
        try:
            ks = C.c_int(0)
            with kernel_profile() as prof:
                call('sf_linear_slabs_fwd', ptr(x), K1, ptr(w), K1, ptr(h), K2, ptr(u), K2, M, N, C.byref(ks),
                     *ws_args(x.device))
            torch.cuda.synchronize()
        finally:
            lib.sf_debug_gate_product_f32(0)
        names = ' '.join(prof.rows)
        assert ('gemm_nt_split_kernel' in names) == (f32 == 0), names
        slabs = workspace(x.device)[:ks.value * M * N * 4].view(torch.float32).view(ks.value, M, N)
        err = slabs.sum(0).double() - ref
        out[f32] = (float(err.abs().max()), float((err.abs() / mag).max()), float(err.pow(2).mean().sqrt()))
    print('[gate product M=%d] fp32 MFMA: max %.2e rel %.2e rms %.2e | bf16 x 6: max %.2e rel %.2e rms %.2e'
          % (M, *out[1], *out[0]))
    assert out[0][1] <= 2.5e-7
    assert out[0][2] <= out[1][2] * 1.05 and out[0][0] <= out[1][0] * 1.25
