"""optim.FusedAdam (sf_adam_step) against torch.optim.Adam: same update rule as train.py:263-268."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g) for s in shapes]


@pytest.mark.parametrize('wd', [0.0, 5e-4])
def test_fused_adam_matches_torch_adam(wd):
    from speaker_follower_amd import optim
    shapes = [(2048, 300), (2048,), (512, 1024), (7,), (3, 5), (1,)]       # odd sizes: the n % 4 tail
    init = _params(0, shapes)
    ref = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    got = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    o_ref = torch.optim.Adam(ref, lr=1e-3, weight_decay=wd)
    o_got = optim.FusedAdam(got, lr=1e-3, weight_decay=wd)
    for it in range(5):
        grads = _params(100 + it, shapes)
        for p, q, g in zip(ref, got, grads):
            p.grad = g.clone().cuda()
            if q.grad is None:
                q.grad = g.clone().cuda()
            else:
                q.grad.copy_(g.cuda())
        v_before = [q._version for q in got]
        o_ref.step()
        o_got.step()
        assert all(q._version > v for q, v in zip(got, v_before))           # caches keyed on _version refresh
        for p, q in zip(ref, got):
            np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    for p, q in zip(ref, got):
        m, v, step = o_got.moments(q)
        st = o_ref.state[p]
        assert step == 5
        np.testing.assert_allclose(m.cpu().numpy(), st['exp_avg'].cpu().numpy(), rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(v.cpu().numpy(), st['exp_avg_sq'].cpu().numpy(), rtol=1e-5, atol=1e-8)


def test_fused_adam_uses_flat_grads_of_dp():
    """With dp.FlatGrads in place the optimizer steps on that buffer (no second copy of the gradients)."""
    from speaker_follower_amd import optim, dp
    a = [torch.nn.Parameter(torch.randn(33, 8).cuda()), torch.nn.Parameter(torch.randn(5).cuda())]
    b = [torch.nn.Parameter(torch.randn(12, 4).cuda())]
    flat = dp.FlatGrads(a + b)
    oa, ob = optim.FusedAdam(a, lr=1e-2), optim.FusedAdam(b, lr=1e-2)
    flat.flat.normal_()
    before = [p.detach().clone() for p in a + b]
    oa.step()
    ob.step()
    assert oa._flat[0]['g'].untyped_storage().data_ptr() == flat.flat.untyped_storage().data_ptr()
    assert ob._flat[0]['g'].data_ptr() == b[0].grad.data_ptr()
    assert flat.attached()
    for p, q in zip(a + b, before):
        assert not torch.equal(p.detach(), q)
    oa.zero_grad()
    assert float(a[0].grad.abs().sum()) == 0.0 and float(b[0].grad.abs().sum()) != 0.0
