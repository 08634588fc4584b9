"""Data parallelism for follower / speaker training: one process per GPU, RCCL over xGMI
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

The R2R minibatch is sharded by rows; samples never interact in the forward or backward
pass except through the per-step loss normaliser (CrossEntropyLoss averages over the
non-ignored rows of the WHOLE batch, follower.py:278,481).  So exactly two exchanges exist:

  1. `allreduce_step_counts`: the [steps, 2] (sum of CE terms, live-row count) table -- a few
     hundred bytes -- so every rank scales its loss by the global denominator;
  2. the gradient sum between backward() and Adam.step() (follower.py:1014-1018) over ONE flat fp32
     buffer (56 MB for the follower's 14.06 M trainable parameters; xGMI is point-to-point, so a few
     large messages that RCCL splits over all 7 links beat many per-tensor calls):
       * `FlatGrads.allreduce`     -- one blocking all-reduce of the whole buffer, or
       * `BucketedGrads`           -- the buffer laid out in PRODUCTION order of the backward (decoder
         LSTM weights first: 40 MB, formed by the first products of sf_attn_decoder_wgrad; the other
         decoder weights; the encoder last) and each bucket's all-reduce launched (async) from the
         backward as soon as the launches that complete it are issued, so that the 40 MB bucket
         travels while the rest of the weight gradients and the encoder's backward through time
         (0.4 ms of dependent launches) still run; `wait()` before the optimizer.

Gradients are SUMMED, not averaged: each rank's loss already carries the global normaliser.
"""
import torch
import torch.distributed as dist

# Test switch (bench.py --force-collectives, tests/rccl_worker.py): issue every collective even in a process group of
# ONE rank.  A one-rank run cannot give a scaling curve, but it executes the real RCCL path on the real GPU -- library
# load, `init_process_group('nccl', device_id=...)`, the stream semantics of async all-reduces launched from the
# backward and waited for before the optimizer -- and its results must equal the no-collective run bit for bit.
FORCE_COLLECTIVES = False


def collectives_on(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return FORCE_COLLECTIVES or dist.get_world_size(group) > 1


def shard_rows(n_rows, rank, world):
    """Contiguous row range of `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


def allreduce_step_counts(sum_cnt, group=None):
    """In-place sum over ranks of the [steps, 2] (term sum, live count) table."""
    if collectives_on(group):
        dist.all_reduce(sum_cnt, op=dist.ReduceOp.SUM, group=group)
    return sum_cnt


def step_losses(sum_cnt):
    """loss = sum_t sum_t / cnt_t (0 where cnt_t == 0); gscale_t = 1 / cnt_t (0 where empty)."""
    s, c = sum_cnt[:, 0], sum_cnt[:, 1]
    live = c > 0
    safe = torch.where(live, c, torch.ones_like(c))
    return torch.where(live, s / safe, torch.zeros_like(s)).sum(), torch.where(
        live, 1.0 / safe, torch.zeros_like(c))


class FlatGrads:
    """One contiguous fp32 gradient buffer; every trainable parameter's .grad is a view of it.

    The HIP backward accumulates weight gradients in place into param.grad
    (runtime.grad_ptr), i.e. straight into this buffer, so the all-reduce needs no packing.
    Use optimizer.zero_grad(set_to_none=False) (or FlatGrads.zero()) to keep the views."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def attached(self):
        base = self.flat.data_ptr()
        end = base + self.flat.numel() * 4
        return all(p.grad is not None and base <= p.grad.data_ptr() < end for p in self.params)

    def zero(self):
        self.flat.zero_()

    def allreduce(self, group=None):
        if not self.attached():
            raise RuntimeError('a parameter .grad was replaced (zero_grad(set_to_none=True)?); '
                               'FlatGrads views are gone')
        if collectives_on(group):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        return self.flat


class BucketedGrads(FlatGrads):
    """FlatGrads whose buffer is a sequence of buckets in the order the backward completes them.

    `launch(b)` must be called with the CURRENT stream being the one the launches that complete
    bucket b were issued on: torch's process group orders the collective behind that stream's work
    and runs it on its own communication stream; `wait()` orders the current stream behind every
    collective launched since the last wait (call it before optimizer.step()).  World size 1: both
    are no-ops."""

    def __init__(self, buckets, group=None):
        buckets = [[p for p in b if p.requires_grad] for b in buckets]
        buckets = [b for b in buckets if b]
        super().__init__([p for b in buckets for p in b])
        self.group = group
        self.bounds, off = [], 0
        for b in buckets:
            n = sum(p.numel() for p in b)
            self.bounds.append((off, off + n))
            off += n
        self._works = []
        self.launched = []
        # runtime.TrainingGraph (segmented capture): a callable that takes the HOST action of a collective point
        # -- start bucket b's all-reduce, wait for all of them -- instead of it being run on the spot: the graph is cut
        # there and the action becomes the step between two replayed segments
        self.hook = None

    def _do(self, action):
        if self.hook is not None:
            self.hook(action)
        else:
            action()

    @property
    def n_buckets(self):
        return len(self.bounds)

    def _active(self):
        return collectives_on(self.group)

    def launch(self, b):
        if b in self.launched:
            raise RuntimeError('bucket %d launched twice before wait()' % b)
        # an optimizer (or zero_grad(set_to_none=True)) that re-bound a .grad would leave this buffer dead: the
        # all-reduce would then sum memory nobody writes and the real gradients would never be reduced
        if not self.attached():
            raise RuntimeError('a parameter .grad no longer lives in the bucket buffer (re-bound by an optimizer or '
                               'zero_grad(set_to_none=True)); gradients would silently not be reduced')
        self.launched.append(b)
        if self._active():
            self._do(lambda: self._start(b))

    def _start(self, b):
        lo, hi = self.bounds[b]
        self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _wait_works(self):
        for w in self._works:
            w.wait()
        self._works = []

    def wait(self):
        if sorted(self.launched) != list(range(self.n_buckets)):
            missing = sorted(set(range(self.n_buckets)) - set(self.launched))
            # refuse -- but never with a collective still in flight: whoever catches this owns a QUIESCENT buffer
            # (every rank launched the same buckets, so the waits complete)
            self.abort()
            raise RuntimeError('gradient buckets %s were never launched: the backward did not run to the end' % missing)
        self.launched = []
        if self._active():
            self._do(self._wait_works)
        return self.flat

    def abort(self):
        """Gives up the current round of buckets (FollowerEngine.run re-issuing a faulted pass): waits for the
        all-reduces already in flight -- every rank launched the same ones, the decision to re-issue is taken on a
        reduced flag -- so that the buffer may be zeroed and the buckets launched again."""
        for w in self._works:
            w.wait()
        self._works, self.launched = [], []


def follower_buckets(encoder, decoder):
    """The follower's trainable parameters in the order FollowerEngine._backward completes their gradients:
    [decoder LSTM (weight_ih 35.7 MB, weight_hh, biases)], [the other decoder weights], [encoder]."""
    lstm = list(decoder.lstm.parameters())
    ids = {id(p) for p in lstm}
    rest = [p for p in decoder.parameters() if id(p) not in ids]
    return [lstm, rest, list(encoder.parameters())]


def allreduce_gradients(params, group=None):
    """Fallback for parameters whose grads are not in a FlatGrads buffer: pack, reduce, unpack."""
    ps = [p for p in params if p.requires_grad and p.grad is not None]
    if not ps or not collectives_on(group):
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in ps:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p))
        off += n
