"""Host-side plumbing between torch tensors and the C ABI: raw pointers, the current
HIP stream, the scratch workspace, and the pointer structs of include/sf_hip.h.

PyTorch is used for device memory and streams only; every arithmetic op on the hot
path goes through libsf_hip.so.  There is no CPU implementation: tensors that are
not on a GPU raise.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check  # noqa: F401

_workspaces = {}


def gc_paused(fn):
    """Decorator for the host procedures that build ~10^4-10^5 small containers per call (search hypotheses and
    result dictionaries, the speaker's per-route outputs): every 700 container allocations the cyclic collector
    would start walking the process' object graph -- with 90 parsed scans that is tens of milliseconds a pass,
    several times the procedure itself (measured: 22 ms for a 2 500-row output loop that takes 6).  Nothing these
    procedures build is cyclic garbage: the collector is paused for the duration and left as it was found."""
    import functools
    import gc

    @functools.wraps(fn)
    def run(*a, **k):
        was = gc.isenabled()
        gc.disable()
        try:
            return fn(*a, **k)
        finally:
            if was:
                gc.enable()
    return run


import contextlib


@contextlib.contextmanager
def graph_capture(graph, stream_):
    """`torch.cuda.graph(graph, stream=...)` with the cyclic collector OFF for the duration (torch collects once before
    it starts).  A collection that happens to run inside a capture -- on this thread or on autograd's worker -- may
    finalise an older hipGraph, stream or event left in a reference cycle; destroying those is not permitted while a
    stream captures, the error is raised inside a destructor and the process aborts (seen once the agents' loops
    shifted where the collector's thresholds fall: tests/test_gpu_agents.py under round 5's prefetching)."""
    import gc
    was = gc.isenabled()
    try:
        with torch.cuda.graph(graph, stream=stream_):
            gc.disable()                                 # (torch's own gc.collect() has just run)
            yield
    finally:
        if was:
            gc.enable()


_concurrent = {}


def concurrent_stream(device, exclude=()):
    """A torch stream whose kernels really run BESIDE the current stream's.

    HIP multiplexes streams onto a handful of hardware queues; two streams that land on the same queue execute in
    order, whatever the program says.  Measured on MI355X (tools/stream_queue_probe.py): the training iteration takes
    5.05-5.17 ms with the backward's side stream on most pool streams and 5.63 ms on one of them -- which one an engine
    gets depends on how many streams the process created before.  So a candidate is PROBED: a spin kernel on the
    current stream and one on the candidate, started together; if the pair takes much more than one of them, the
    candidate shares the queue and the next pool stream is tried.  Once per (device, current stream), ~2 ms."""
    import time
    idx = device.index if device.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(device)
    key = (idx, cur.cuda_stream)
    exclude = [x for x in exclude if x is not None]
    found = _concurrent.setdefault((key, tuple(x.cuda_stream for x in exclude)), [])
    if found:
        return found[0]
    if not hasattr(torch.cuda, '_sleep') or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device=device)
    spin = 400000                                       # ~0.2 ms of device cycles

    def timed(streams):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(spin)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0
    timed([cur])                                         # (first launch of the spin kernel)
    one = min(timed([cur]) for _ in range(2))
    cand = None
    for _ in range(8):
        cand = torch.cuda.Stream(device=device)
        if any(cand.cuda_stream == x.cuda_stream for x in exclude):
            continue
        group = [cur] + exclude + [cand]                 # (concurrent with the current stream AND with the excluded ones)
        timed(group)
        if min(timed(group) for _ in range(2)) < 1.5 * one:
            found.append(cand)
            return cand
    return cand                                          # (none ran concurrently: e.g. one hardware queue)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                'speaker_follower_amd: the hot path runs on MI355X through libsf_hip.so only; '
                'got a %s tensor (there is no CPU fallback)' % t.device)


def ptr(t):
    """Device pointer of a contiguous fp32/int tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError('non-contiguous tensor passed to the C ABI')
    return C.c_void_p(t.data_ptr())


def f32(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError('expected float32, got %s' % t.dtype)
    return t.contiguous()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(device):
    """One scratch buffer per (device, stream): split-K slabs and backward temporaries."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            # a zero-fill issued here would become a node of the graph: 64 MB re-zeroed by EVERY replay (15-30 us; it
            # rode in every captured rollout until round 5: profiles/r05_z_*)
            raise RuntimeError('the workspace of a capture stream must exist before the capture begins: '
                               'runtime.ensure_workspace(stream, device)')
        ws = torch.zeros(lib.sf_workspace_bytes(), dtype=torch.uint8, device=device)   # zero ONCE: sf_hip.h
        _workspaces[key] = ws
    return ws


def ensure_workspace(stream_, device):
    """Creates (and zero-fills, eagerly) the workspace of `stream_` -- what every capture does for its capture stream
    BEFORE torch.cuda.graph(...): a workspace first touched inside a capture would be zero-filled by every replay."""
    with torch.cuda.stream(stream_):
        workspace(device)
    stream_.synchronize()


_wgrad_workspaces = {}


def wgrad_ws_args(device):
    """(pointer, bytes, stream) of the larger scratch buffer of the weight-gradient entry points (sf_hip.h:
    sf_wgrad_workspace_bytes), one per (device, stream), allocated on first use -- inference never touches it."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream().cuda_stream)
    ws = _wgrad_workspaces.get(key)
    if ws is None:
        ws = _wgrad_workspaces[key] = torch.empty(lib.sf_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
    return C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), stream()


def ws_args(device):
    ws = workspace(device)
    return C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), stream()


# ---- strict summation order of the gate products (include/sf_hip.h: sf_gate_product_strict) -----------------
class strict_gate_product:
    """`with runtime.strict_gate_product():` -- the LSTM gate products on the fp32 MFMA (the summation order with which
    the state-factored search reproduces the reference's traversal at exact fp32 ties too) instead of the bf16x6 split
    products.  Process-wide switch; restores the previous setting on exit.  Captured graphs keep the kernels they were
    captured with: capture inside the block what is to replay strictly."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = int(lib.sf_gate_product_is_strict())
        lib.sf_gate_product_strict(int(self.on))
        return self

    def __exit__(self, *exc):
        lib.sf_gate_product_strict(self.prev)
        return False


# ---- the fault word of the persistent launches (include/sf_hip.h: sf_workspace_fault_offset) ----------------
FAULT_ENC_FWD, FAULT_ENC_BWD, FAULT_SPEAKER, FAULT_LOCK = 1, 2, 4, 8


class WeightsMoved(RuntimeError):
    """A captured pass refuses to replay: a weight tensor (or one of its cached derived layouts) was re-allocated since
    the capture, and an address cannot be patched into a graph.  Callers that can re-capture catch THIS type -- a
    HIP launch or graph error surfaced by torch is a plain RuntimeError and must not be mistaken for it."""


class PersistentLaunchFault(RuntimeError):
    """A persistent launch gave up a bounded wait (its outputs are NaN-poisoned) and the per-step re-issue
    failed as well, or no re-issue was possible."""


def _fault_view(ws):
    off = lib.sf_workspace_fault_offset(ws.numel())
    return ws[off:off + 4].view(torch.int32)


def fault_word(device):
    """The fault word (int32[1] view) of the CURRENT stream's workspace: what the persistent launches issued on this
    stream raise, what optim.FusedAdam.guard_faults makes a device-side optimizer step look at."""
    return _fault_view(workspace(device))


def fault_views(device):
    """The fault words (int32[1] views) of every workspace of `device`: for callers that read them with their own
    asynchronous copy (`torch.cat(views)` into pinned memory) instead of take_fault's synchronous one, and zero the
    ones they found raised."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return [_fault_view(ws) for (d, _), ws in _workspaces.items() if d == idx]


def take_fault(device):
    """Reads AND clears the fault words of EVERY workspace of `device` (one per stream that ever ran library
    calls: the backward's side streams and captured rollouts have their own).  A host sync: callers put it where
    they synchronise anyway (the D2H copy of a rollout's actions, the loss read of a training iteration).  Returns
    the OR of the FAULT_* bits raised since the last call; 0 in a healthy process."""
    views = fault_views(device)
    if not views:
        return 0
    words = (torch.cat(views) if len(views) > 1 else views[0]).cpu()        # (the sync)
    bits = 0
    for w, b in zip(views, words.tolist()):
        if b:
            w.zero_()
            bits |= b
    return bits


def dropout_arg(p, seed, row0=0, site_dev=None, site_mul=1):
    """sf_dropout* (NULL when p == 0: eval mode).  site_dev: device word added (x site_mul) to every stream id of this
    configuration -- the site counter of a captured training iteration (include/sf_hip.h: sf_dropout.site_dev)."""
    if not p:
        return None
    return C.byref(_lib.Dropout(float(p), int(seed) & 0xFFFFFFFF, int(row0), site_dev, int(site_mul)))


def site_word(device, value=0):
    """A device-side site counter (one uint32 word, held as int32) for sf_dropout.site_dev / sf_sample.stream_dev."""
    return torch.full((1,), int(value), dtype=torch.int32, device=device)


def site_advance(word, by):
    _lib.call('sf_site_advance', C.c_void_p(word.data_ptr()), int(by), stream())


def fill_regions(*pairs):
    """sf_fill_regions: contiguous tensors set to a constant each in one launch -- `fill_regions((hs[0], 0.0), (words[0],
    BOS), (ended, 0))` instead of one torch fill kernel per tensor.  Up to 8 (tensor, value) pairs."""
    import struct
    regs = (_lib.FillRegion * len(pairs))()
    for r, (x, v) in zip(regs, pairs):
        assert x.is_contiguous() and x.is_cuda
        width = x.element_size()
        if x.dtype == torch.float32:
            v = struct.unpack('<I', struct.pack('<f', float(v)))[0]
        elif x.dtype in (torch.int64, torch.int32, torch.uint8, torch.int8, torch.bool):
            v = int(v) & ((1 << (8 * width)) - 1)
        else:
            raise TypeError('fill_regions: %s' % x.dtype)
        r.ptr, r.count, r.value, r.width = x.data_ptr(), x.numel(), v, width
    _lib.call('sf_fill_regions', regs, len(pairs), stream())


class TrainingGraph:
    """ONE whole training iteration -- zero the gradients, rollout / scoring pass in train mode, backward through
    time, optimizer steps -- as a hipGraph (follower.py:1001-1020, speaker.py:376-395 per replay).

    Kernel arguments are frozen in a graph; what must differ between two iterations lives in device memory instead:
    the dropout / sampling SITE counter (sf_dropout.site_dev, sf_sample.stream_dev) and the optimizers' step counters
    (sf_adam_step_dev).  `replay()` writes those four words with one tiny launch (values travel as kernel arguments of
    THAT launch) and launches the graph: iteration n of a replayed loop draws exactly the masks, samples and bias
    corrections of iteration n of the eager loop (tests/test_gpu_training_graph.py).

    `body()` must issue one complete iteration on the current stream and return the pass state; it is run ONCE
    eagerly (a real training step: caches, workspaces, streams; its state is `.first`) and then captured (`.state`: the
    tensors every replay overwrites).  `engine` carries `site_next`
    (host mirror of the counter), `iteration` and `site_word`."""

    def __init__(self, engine, body, optimizers, device, segmented=None):
        """segmented: None, or the objects (besides `engine`) whose collective points cut the capture -- each has a
        `.hook` attribute (dp.BucketedGrads); the engine's is `collective_hook`.  A replay is then a SEQUENCE of graph
        launches with the host actions of the collective points between them (FollowerEngine._capture_training_dp)."""
        self.engine, self.optimizers = engine, list(optimizers)
        groups = sum(len(o.live_groups()) for o in self.optimizers)
        if groups > 3:
            raise ValueError('at most three optimizer parameter groups per training graph')
        self.ctl = torch.zeros(4, dtype=torch.int32, device=device)            # [site, step, step, step]
        k = 1
        for o in self.optimizers:
            n = len(o.live_groups())
            o.bind_device_steps([self.ctl[k + i:k + i + 1] for i in range(n)])
            k += n
        # every device block whose ADDRESS the captured launches carry lives as long as the graph does
        self._scratch = [t for o in self.optimizers for t in o.device_scratch()]
        engine.site_word = self.ctl[0:1]
        self.stream = torch.cuda.Stream(device=device)
        cur = torch.cuda.current_stream(device)
        self.stream.wait_stream(cur)
        self.segments = None
        hooked = ()
        try:
            with torch.cuda.stream(self.stream):
                self._store()
                self.first = body()                     # eager: a real iteration, on the capture stream (its state)
                torch.cuda.synchronize(device)
                invalidate_caches()                     # every derived weight copy is refreshed INSIDE the graph
                keep = (engine.site_next, engine.iteration, [o.host_steps() for o in self.optimizers])
                if segmented is None:
                    self.graph = torch.cuda.CUDAGraph()
                    with graph_capture(self.graph, self.stream):
                        self.state = body()
                else:
                    hooked = tuple(segmented)
                    self._capture_segments(engine, hooked, body, device)
                # the capture advanced the host mirrors without running anything
                engine.site_next, engine.iteration = keep[0], keep[1]
                for o, st_ in zip(self.optimizers, keep[2]):
                    o.set_host_steps(st_)
            cur.wait_stream(self.stream)
        finally:
            engine.site_word = None
            engine.collective_hook = None
            for h in hooked:
                h.hook = None
            for o in self.optimizers:
                o.bind_device_steps(None)
        self.stride = self.state.site_stride
        self.replays = 0

    def _capture_segments(self, engine, hooked, body, device):
        """Captures `body()` as a chain of graphs cut at its collective points.  At a cut: every stream the body has
        forked from the capture stream and not joined yet (the engine lists them in `_open_forks`) is joined, the
        capture ends, the host action is RECORDED (not run: a capture executes nothing), a new capture begins in the
        same memory pool and the forks are re-opened, so the launches that follow on a side stream land in it."""
        import gc
        origin = self.stream
        pool = torch.cuda.graph_pool_handle()
        self.segments = []
        cur = {'g': None}

        def begin():
            cur['g'] = torch.cuda.CUDAGraph()
            # (thread_local: a process group's watchdog thread may touch the runtime while this thread captures)
            cur['g'].capture_begin(pool=pool, capture_error_mode='thread_local')

        def end(action):
            cur['g'].capture_end()
            self.segments.append((cur['g'], action))

        def cut(action):
            here = torch.cuda.current_stream(device)
            forks = [s_ for s_ in list(getattr(engine, '_open_forks', ())) + [here] if s_ != origin]
            for s_ in forks:
                origin.wait_stream(s_)
            with torch.cuda.stream(origin):
                end(action)
                begin()
            for s_ in forks:
                s_.wait_stream(origin)

        torch.cuda.synchronize(device)
        gc.collect()
        was = gc.isenabled()
        gc.disable()                                     # (see graph_capture: no finaliser may run inside a capture)
        engine.collective_hook = cut
        for h in hooked:
            h.hook = cut
        try:
            begin()
            try:
                self.state = body()
            except BaseException:
                # never leave the stream capturing: end the open segment (whatever it holds is dropped with the graphs)
                try:
                    for s_ in list(getattr(engine, '_open_forks', ())):
                        origin.wait_stream(s_)
                    with torch.cuda.stream(origin):
                        cur['g'].capture_end()
                except Exception:                            # noqa: BLE001  (the original error is the one to report)
                    pass
                self.segments = None
                raise
            end(None)
        finally:
            if was:
                gc.enable()
        self.graph = None

    def _store(self):
        steps = [s for o in self.optimizers for s in o.host_steps()] + [0, 0, 0]
        _lib.call('sf_store_u32x4', C.c_void_p(self.ctl.data_ptr()), int(self.engine.site_next) & 0xFFFFFFFF,
                  int(steps[0]), int(steps[1]), int(steps[2]), stream())

    def replay(self):
        """One training iteration on the current stream.  Returns the (static) pass state: its tensors are
        overwritten by every replay; `state.site0` is the host mirror of the sites this replay uses."""
        eng = self.engine
        self._store()
        if self.segments is None:
            self.graph.replay()
        else:
            for g, action in self.segments:              # (one stream: every host action is ordered behind its segment)
                g.replay()
                if action is not None:
                    action()
        self.state.site0 = eng.site_next
        eng.site_next += self.stride
        eng.iteration += 1
        for o in self.optimizers:
            o.set_host_steps([s + 1 for s in o.host_steps()])
            o.bump_versions()                           # eager code that follows must rebuild its derived weight copies
        self.replays += 1
        return self.state


def grad_ptr(param):
    """Pointer to the in-place gradient accumulator of a parameter (NULL if frozen).

    Weight gradients are accumulated by the kernels directly into `param.grad`
    (allocated zero-filled on first use), which is also the buffer the
    data-parallel all-reduce works on -- no per-step temporaries."""
    if param is None or not param.requires_grad:
        return None
    if param.grad is None:
        param.grad = torch.zeros_like(param, memory_format=torch.contiguous_format)
    return C.c_void_p(param.grad.data_ptr())


def lstm_w(w_ih, w_hh, b_ih, b_hh, grad=False):
    g = grad_ptr if grad else ptr
    return _lib.LstmW(*(_v(g(t)) for t in (w_ih, w_hh, b_ih, b_hh)))


def _v(p):
    return p.value if p is not None else None


def struct_of(cls, tensors, grad=False):
    g = grad_ptr if grad else ptr
    return cls(*(_v(g(t)) for t in tensors))


def pano_dense(X):
    B, V, F = X.shape
    return _lib.Pano(X.data_ptr(), None, None, None, None, V, F, 0)


def cands_dense(U):
    B, A, F = U.shape
    return _lib.Cands(U.data_ptr(), None, None, None, None, None, A, 1, F, 0)


_transposed = {}
_cache_epoch = 0


def invalidate_caches():
    """Forces every derived copy of the weights (transposed layouts, the [vocab,4H] input-product
    tables, the folded inference matrices) to be rebuilt on next use.

    The caches are keyed on `(param.data_ptr(), param._version)`: every in-place update torch knows
    about (optimizer.step(), `load_state_dict`, `p.add_()` under no_grad, FusedAdam) bumps `_version`
    and is noticed by itself.  A write through `param.data` (or a raw pointer) is NOT -- `.data`
    carries its own version counter -- so code that does that must call this afterwards."""
    global _cache_epoch
    _cache_epoch += 1


def cache_epoch():
    return _cache_epoch


def weight_key(*tensors):
    """Cache key of a set of weights: storage, torch version counter and the invalidation epoch."""
    return tuple((t.data_ptr(), t._version) for t in tensors) + (_cache_epoch,)


def xw_table(owner, emb, w_ih):
    """[vocab, 4H] = embedding W_ih^T kept on `owner` (a module) and rebuilt IN PLACE when either
    tensor changes: an LSTM whose input is an embedding lookup reads its input product as a row of
    this table.  In place, so that a captured hipGraph keeps pointing at live memory (see
    FollowerEngine.capture)."""
    key = weight_key(emb, w_ih)
    if getattr(owner, '_xw_key', None) != key:
        V, N = emb.shape[0], w_ih.shape[0]
        buf = getattr(owner, '_xw', None)
        if buf is None or buf.shape != (V, N) or buf.device != emb.device:
            buf = torch.empty(V, N, device=emb.device, dtype=torch.float32)
        e, w = emb.detach().contiguous(), w_ih.detach().contiguous()
        _lib.call('sf_linear_fwd', ptr(e), e.shape[1], ptr(w), None, V, N, e.shape[1], 0, ptr(buf), N,
                  *ws_args(emb.device))
        owner._xw, owner._xw_key = buf, key
    return owner._xw


_query_folds = {}


def visual_query_fold(w_h, b_h, w_v):
    """sf_visual_fold64 of a VisualSoftDotAttention (model.py:303-316): M_v = W_v^T W_h [F,H] and c_v = W_v^T b_h [F] as
    float64 device tensors, rebuilt (in place) only when one of the three weights changed.  Returns (struct, keep-alive)."""
    import weakref
    key = (w_h.data_ptr(), b_h.data_ptr(), w_v.data_ptr())
    ver = (w_h._version, b_h._version, w_v._version, _cache_epoch)
    hit = _query_folds.get(key)
    if hit is not None and hit[0]() is w_h and hit[1] == ver:
        return hit[2], hit[3]
    D, H = w_h.shape
    F = w_v.shape[1]
    if hit is not None and hit[0]() is w_h:
        m_v, c_v = hit[3]
    else:
        m_v = torch.empty(F, H, device=w_h.device, dtype=torch.float64)
        c_v = torch.empty(F, device=w_h.device, dtype=torch.float64)
    vw = _lib.VisualW(w_h.data_ptr(), b_h.data_ptr(), w_v.data_ptr(), None, transposed(w_v).data_ptr(),
                      transposed(w_h).data_ptr())
    _lib.call('sf_visual_query_fold_f64', C.byref(vw), H, D, F, C.c_void_p(m_v.data_ptr()), C.c_void_p(c_v.data_ptr()), stream())
    fold = _lib.VisualFold64(m_v.data_ptr(), c_v.data_ptr())
    if len(_query_folds) > 64:
        for k in [k for k, v in _query_folds.items() if v[0]() is None]:
            del _query_folds[k]
    _query_folds[key] = (weakref.ref(w_h), ver, fold, (m_v, c_v))
    return fold, (m_v, c_v)


def transposed(w):
    """Device copy of w^T for a 2-D weight, rebuilt (in place) only when the weight changed (torch
    bumps `_version` on every in-place update, e.g. optimizer.step(); see invalidate_caches for
    the one case it cannot see).  288 GB of HBM make keeping every hot weight in both layouts free;
    it turns y = x W into a K-contiguous product."""
    import weakref
    key = w.data_ptr()
    hit = _transposed.get(key)
    if hit is not None and hit[0]() is w and hit[1] == (w._version, _cache_epoch):
        return hit[2]
    R, Cc = w.shape
    out = hit[2] if (hit is not None and hit[2].shape == (Cc, R) and hit[2].device == w.device) \
        else torch.empty(Cc, R, device=w.device, dtype=torch.float32)
    _lib.call('sf_transpose', ptr(w.detach()), R, Cc, ptr(out), stream())
    if len(_transposed) > 256:                      # dead entries of freed modules
        for k in [k for k, v in _transposed.items() if v[0]() is None]:
            del _transposed[k]
    _transposed[key] = (weakref.ref(w), (w._version, _cache_epoch), out)
    return out
