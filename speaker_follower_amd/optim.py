"""The optimizer step of the training loop on the HIP path.

train.py:263-268 builds `optim.Adam(params, lr=1e-4, weight_decay=5e-4)` twice (encoder,
decoder) and the agents call `.step()` after `loss.backward()` (follower.py:1014-1020,
speaker.py:389-395).  `FusedAdam` is a drop-in for that class: same constructor arguments, same
update rule (L2 weight decay added to the gradient, bias-corrected moments, no amsgrad), but the
parameters of the optimizer live in ONE flat fp32 buffer (every `param.data` becomes a view of it),
so do the two moments, and a step is ONE launch of `sf_adam_step` instead of torch's chain of
multi-tensor kernels (0.4 ms of an 8 ms training iteration for the follower's 14 M parameters).

Gradients: when `dp.FlatGrads` already holds the parameters' `.grad` as consecutive views of one
buffer (the layout the RCCL all-reduce wants) that buffer is used as it is; otherwise the optimizer
creates its own flat gradient buffer and binds the views.  Keep them: `zero_grad()` zeroes in place.
"""
import torch

from . import _lib
from .runtime import ptr, stream


def _flat_view_of_grads(params):
    """The parameters' .grad as one contiguous range of one storage, or None."""
    g0 = params[0].grad
    if g0 is None or g0.dtype != torch.float32 or not g0.is_contiguous():
        return None
    store = g0.untyped_storage()
    off = g0.storage_offset()
    expect = off
    for p in params:
        g = p.grad
        if (g is None or g.dtype != torch.float32 or not g.is_contiguous()
                or g.untyped_storage().data_ptr() != store.data_ptr() or g.storage_offset() != expect):
            return None
        expect += g.numel()
    return torch.empty(0, dtype=torch.float32, device=g0.device).set_(store, off, (expect - off,), (1,))


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) with one HIP launch per step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            if any(p.dtype != torch.float32 or not p.is_cuda for p in ps):
                raise ValueError('FusedAdam runs on the HIP path: float32 parameters on the GPU only '
                                 '(there is no CPU fallback)')
            dev = ps[0].device
            total = sum(p.numel() for p in ps)
            flat_p = torch.empty(total, dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for p in ps:
                    n = p.numel()
                    flat_p[off:off + n].copy_(p.data.reshape(-1))
                    p.data = flat_p[off:off + n].view_as(p)      # the parameter now lives in the flat buffer
                    off += n
            self._flat.append(dict(params=ps, p=flat_p, g=None, step=0, step_dev=None, coef=None,
                                   m=torch.zeros(total, dtype=torch.float32, device=dev),
                                   v=torch.zeros(total, dtype=torch.float32, device=dev)))

    def _grads(self, f):
        g = f['g']
        if g is not None and all(p.grad is not None and p.grad.untyped_storage().data_ptr() ==
                                 g.untyped_storage().data_ptr() for p in f['params']):
            return g
        g = _flat_view_of_grads(f['params'])
        if g is None:                      # bind our own flat gradient buffer (keeps what is there)
            g = torch.zeros_like(f['p'])
            off = 0
            for p in f['params']:
                n = p.numel()
                if p.grad is not None:
                    g[off:off + n].copy_(p.grad.reshape(-1))
                p.grad = g[off:off + n].view_as(p)
                off += n
        f['g'] = g
        return g

    def zero_grad(self, set_to_none=False):
        """Zeroes in place (the flat views must survive); set_to_none is accepted and ignored."""
        for f in self._flat:
            if f is not None:
                self._grads(f).zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            g = self._grads(f)
            f['step'] += 1
            b1, b2 = group['betas']
            if f['step_dev'] is not None:
                # the step counter lives on the device (bind_device_steps): the kernel increments and uses it
                _lib.call('sf_adam_step_dev', ptr(f['p']), ptr(g), ptr(f['m']), ptr(f['v']), f['p'].numel(),
                          float(group['lr']), float(b1), float(b2), float(group['eps']),
                          float(group['weight_decay']), f['step_dev'], ptr(f['coef']), self._guard_ptr(f['p'].device),
                          stream())
            else:
                _lib.call('sf_adam_step', ptr(f['p']), ptr(g), ptr(f['m']), ptr(f['v']), f['p'].numel(),
                          float(group['lr']), float(b1), float(b2), float(group['eps']),
                          float(group['weight_decay']), int(f['step']), stream())
            # the kernel wrote behind torch's back: bump the version counters so that everything
            # keyed on `param._version` (transposed weight copies, folded tables) is rebuilt
            ps = f['params']
            torch._C._autograd._unsafe_set_version_counter(ps, [p._version + 1 for p in ps])
        return loss

    # `guard_faults = True`: a device-side step (bind_device_steps) becomes a no-op while the fault word of the CURRENT
    # stream's workspace is set -- a captured iteration never steps on gradients a starved persistent launch poisoned
    guard_faults = False

    def _guard_ptr(self, device):
        if not self.guard_faults:
            return None
        from .runtime import fault_word
        return _lib.C.c_void_p(fault_word(device).data_ptr())

    # ---- a step that can live inside a hipGraph (runtime.TrainingGraph): the 1-based step counter is a device word the
    # host writes in front of every replay (kernel arguments are frozen in a graph, device memory is not)
    def live_groups(self):
        return [f for f in self._flat if f is not None]

    def bind_device_steps(self, words):
        """words: device int32 tensor views, one per live parameter group (None: back to host-side counters)."""
        live = self.live_groups()
        if words is None:
            # back to host-side counters.  The `coef` scratch STAYS allocated for the life of the optimizer: its address
            # is a frozen kernel argument of every graph captured while it was bound, and each replay of such a graph
            # writes and reads it -- handing the block back to the caching allocator here would let any later small
            # tensor land under those writes (advisor, round 5)
            for f in live:
                f['step_dev'] = None
            return
        assert len(words) == len(live)
        for f, w in zip(live, words):
            f['step_dev'] = _lib.C.c_void_p(w.data_ptr())
            if f['coef'] is None:
                f['coef'] = torch.zeros(4, dtype=torch.float32, device=f['p'].device)
            f['step_word'] = w

    def device_scratch(self):
        """Tensors whose addresses are baked into a graph captured under bind_device_steps (a TrainingGraph keeps them)."""
        return [f['coef'] for f in self.live_groups() if f['coef'] is not None]

    def host_steps(self):
        return [f['step'] for f in self.live_groups()]

    def set_host_steps(self, steps):
        for f, s_ in zip(self.live_groups(), steps):
            f['step'] = int(s_)

    def bump_versions(self):
        for f in self.live_groups():
            ps = f['params']
            torch._C._autograd._unsafe_set_version_counter(ps, [p._version + 1 for p in ps])

    # ---- torch.optim.Adam <-> this optimizer (agents.Seq2SeqAgent.train adopts the reference's own `optim.Adam` objects,
    # train.py:263-268, for its replayed iterations and hands the state back afterwards)
    @staticmethod
    def adoptable(opt):
        """A torch.optim.Adam this class reproduces exactly: one parameter group, no amsgrad / maximize."""
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return False
        g = opt.param_groups[0]
        return not (g.get('amsgrad') or g.get('maximize') or g.get('differentiable'))

    @torch.no_grad()
    def load_torch_state(self, opt):
        """Hyper-parameters, moments and step count of `opt` (same parameters, same order) into this optimizer."""
        g = opt.param_groups[0]
        mine = self.param_groups[0]
        mine['lr'], mine['betas'], mine['eps'], mine['weight_decay'] = g['lr'], tuple(g['betas']), g['eps'], g['weight_decay']
        f = self._flat[0]
        off, step = 0, 0
        for p in f['params']:
            n = p.numel()
            st = opt.state.get(p)
            if st:
                f['m'][off:off + n].copy_(st['exp_avg'].reshape(-1))
                f['v'][off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                step = max(step, int(st['step']))
            else:
                f['m'][off:off + n].zero_()
                f['v'][off:off + n].zero_()
            off += n
        f['step'] = step

    @torch.no_grad()
    def store_torch_state(self, opt):
        """The other way: `opt.state` as torch.optim.Adam would hold it after the same steps."""
        f = self._flat[0]
        if f['step'] == 0:
            return
        off = 0
        for p in f['params']:
            n = p.numel()
            st = opt.state[p]
            if 'exp_avg' not in st or st['exp_avg'].shape != p.shape:
                st['exp_avg'], st['exp_avg_sq'] = torch.zeros_like(p), torch.zeros_like(p)
            st['exp_avg'].copy_(f['m'][off:off + n].view_as(p))
            st['exp_avg_sq'].copy_(f['v'][off:off + n].view_as(p))
            old = st.get('step')
            st['step'] = (torch.tensor(float(f['step']), dtype=old.dtype, device=old.device) if torch.is_tensor(old)
                          else torch.tensor(float(f['step'])))
            off += n

    # moments in the layout of torch.optim.Adam's state (per parameter), for inspection / tests
    def moments(self, p):
        for f in self._flat:
            if f is None:
                continue
            off = 0
            for q in f['params']:
                n = q.numel()
                if q is p:
                    return f['m'][off:off + n].view_as(p), f['v'][off:off + n].view_as(p), f['step']
                off += n
        raise KeyError('parameter is not managed by this optimizer')
