"""Shared helpers of the per-sample wrappers (B = 1 batches on the current HIP device).

A per-sample call does not allocate device memory.  One pooled context per (process, host thread, device, stream, H, W, size
bucket, plan flags, pacing) keeps the plan and its workspace, a pinned staging buffer for the events (the window's two
offsets travel in front of them: ONE H2D per call), the device buffers, the output tensors and a pool of pinned result
buffers; a call is  host convert -> async H2D -> bin -> build -> async copy of the window statistics -> async D2H of the
result -> ONE stream synchronisation -> the status word is checked and the exception the reference raises is raised.

**What a wrapper returns is the caller's own (r04).**  "outputs are fresh arrays" -- a caller may keep any number
of results (a list comprehension over samples, a caching dataset, a collate over a batch).  The array handed out is a view of a
pinned pool buffer, and a pool buffer is handed out again only when nothing outside refers to it any more (numpy keeps every
view's ``base`` pointing at the buffer's master array, so the master's reference count says so exactly); while
``RESULT_POOL_DEPTH`` buffers of a shape are all still held by the caller, further results are plain fresh numpy arrays (one
more host copy).  A caller that drops each result before the next call (the reference's resize / ``torch.tensor()`` flows)
keeps the pooled speed; nobody sees a buffer change under their feet.  ``out=`` copies into the caller's own array;
``device_out=True`` (``get_item_transform_cuda``) skips the D2H and returns a fresh ``(H, W, C)`` CUDA tensor.
"""
import ctypes
import os
import sys
import threading

import numpy as np
import torch

from .. import _lib
from ..engine import EventBatch, _ptr, _require_gpu
from ..synthetic import from_structured, narrow_to_int32

RESULT_POOL_DEPTH = int(os.environ.get("EVREP_RESULT_POOL", os.environ.get("EVREP_RESULT_RING", "8")))
_CONTEXTS = {}            # insertion-ordered: least recently used first (_context)
# how many (process, thread, device, stream, H, W, size bucket, flags, pacing) contexts stay cached: EVREP_MAX_CONTEXTS
# (a context a caller still holds results of stays alive through their references)
MAX_CONTEXTS = max(1, int(os.environ.get("EVREP_MAX_CONTEXTS", "64")))
_EVICTION_LOGGED = [False]
_CONTEXTS_LOCK = threading.Lock()


class _ResultSlot:
    """One pinned result buffer.  ``master`` is the numpy array every handed-out view has as its ``base``; the slot is
    free when only this object refers to it."""

    def __init__(self, shape, dtype, pin=True):
        self.tensor = torch.empty(shape, dtype=dtype, pin_memory=pin)
        self.master = self.tensor.numpy()
        self.idle = sys.getrefcount(self.master)        # the references this object itself accounts for

    def free(self):
        return sys.getrefcount(self.master) <= self.idle


class _SampleContext:
    """Everything a B = 1 call needs, allocated once."""

    def __init__(self, height, width, cap, device, plan_flags, pacing):
        self.cap = int(cap)
        self.device = device
        # [offsets: 2 x int64 = one 16-byte row][events: cap rows]: one pinned buffer, one device buffer, one H2D
        self.buf_pinned = torch.zeros((self.cap + 1, 4), dtype=torch.int32, pin_memory=True)
        self.buf_dev = torch.zeros((self.cap + 1, 4), dtype=torch.int32, device=device)
        self.ev_pinned = self.buf_pinned[1:]
        self.ev_dev = self.buf_dev[1:]
        self.off_pinned = self.buf_pinned[0].view(torch.int64)          # (2,) int64
        self.batch = EventBatch(self.ev_dev, torch.tensor([0, self.cap], dtype=torch.int64), height, width,
                                max_events_per_window=self.cap, plan_flags=plan_flags, pacing=pacing)
        self.batch.offsets = self.buf_dev[0].view(torch.int64)          # the device offsets live in front of the events
        # host-side views made once (a torch index / slice / clone costs 2-5 us each; a sample has ~100 us in all)
        self.ev_np = self.ev_pinned.numpy()
        self.off_np = self.off_pinned.numpy()
        self.off_host = torch.zeros(2, dtype=torch.int64)               # the batch's offsets_host, rewritten per sample
        self.off_host_np = self.off_host.numpy()
        self.batch.offsets_host = self.off_host
        # this context is only ever used on the stream it was made for (the stream is part of its key), with that stream's
        # device current: the pooled batch skips the per-call stream look-up and device guard
        self.stream = torch.cuda.current_stream(device)
        self.batch._pinned_stream = ctypes.c_void_p(self.stream.cuda_stream)
        self.meta_pinned = torch.zeros(16, dtype=torch.int32, pin_memory=True)      # one 64-byte WindowMeta
        self.meta_np = self.meta_pinned.numpy()
        self.outs = {}          # (C, dtype) -> (1, H, W, C) device tensor
        self.pools = {}         # (shape, dtype) -> [_ResultSlot]
        self.staging = {}       # (shape, dtype) -> pinned tensor never handed out (results beyond the pool's depth)

    def out(self, C, dtype):
        key = (int(C), dtype)
        if key not in self.outs:
            self.outs[key] = torch.empty((1, self.batch.H, self.batch.W, int(C)), dtype=dtype, device=self.device)
        return self.outs[key]

    def result_slot(self, shape, dtype):
        """A pinned buffer nobody outside refers to, or None when the caller still holds all of this shape's."""
        pool = self.pools.setdefault((tuple(shape), dtype), [])
        for slot in pool:
            if slot.free():
                return slot
        if len(pool) < RESULT_POOL_DEPTH:
            pool.append(_ResultSlot(tuple(shape), dtype))
            return pool[-1]
        return None

    def staging_buffer(self, shape, dtype):
        key = (tuple(shape), dtype)
        if key not in self.staging:
            self.staging[key] = torch.empty(shape, dtype=dtype, pin_memory=True)
        return self.staging[key]


def _context(height, width, n):
    stream = torch.cuda.current_stream()
    dev = stream.device
    cap = 4096
    while cap < n:
        cap *= 2
    flags, pacing = _lib.plan_flags_from_env(), _lib.pacing_from_env()
    # per process (a HIP context does not survive fork()), per host thread and per stream: two threads or two streams never
    # share a staging buffer, an output tensor or a workspace
    key = (os.getpid(), threading.get_ident(), dev.index, int(stream.cuda_stream), int(height), int(width), cap, flags, pacing)
    with _CONTEXTS_LOCK:
        ctx = _CONTEXTS.pop(key, None)
        if ctx is not None:
            _CONTEXTS[key] = ctx            # most recently used last
    if ctx is None:
        ctx = _SampleContext(height, width, cap, dev, flags, pacing)
        with _CONTEXTS_LOCK:
            _CONTEXTS[key] = ctx
            # bounded (r05): contexts of threads that have ended go first, then the least recently used ones -- an application
            # that makes a thread or a stream per sample would otherwise pin cap x 16 B and hold 29.5 MB of device output each
            if len(_CONTEXTS) > MAX_CONTEXTS:
                alive = {t.ident for t in threading.enumerate()}
                for k in [k for k in _CONTEXTS if k[0] != os.getpid() or k[1] not in alive]:    # dead threads, forked-away parents
                    del _CONTEXTS[k]
                # only then live ones, least recently used first -- and never silently: every miss after this re-allocates
                # pinned staging, a workspace and a plan (milliseconds), so a feed that cycles through more shapes / size
                # buckets than the limit should raise EVREP_MAX_CONTEXTS
                while len(_CONTEXTS) > MAX_CONTEXTS:
                    victim = next(k for k in _CONTEXTS if k != key)
                    del _CONTEXTS[victim]
                    if not _EVICTION_LOGGED[0]:
                        _EVICTION_LOGGED[0] = True
                        import warnings
                        warnings.warn("evrep: more than %d live per-sample contexts (threads x streams x sensor sizes x size "
                                      "buckets); the least recently used one was evicted -- set EVREP_MAX_CONTEXTS higher if "
                                      "this feed cycles through that many" % MAX_CONTEXTS, RuntimeWarning, stacklevel=3)
    return ctx


class UnsortedWindow(NotImplementedError):
    """A window whose timestamps are not ascending reached a wrapper that needs them so (or needs to take its slow path)."""


class SampleBatch:
    """One pooled window: ``.batch`` is the resident EventBatch (its builders take ``out=``), ``.out(C, dtype)`` the pooled
    output tensor, ``finish()`` the one synchronisation."""

    def __init__(self, ctx, n, device_out=False):
        self.ctx, self.batch, self.n = ctx, ctx.batch, int(n)
        self.H, self.W = ctx.batch.H, ctx.batch.W
        self.device_out = bool(device_out)

    def out(self, C, dtype):
        if self.device_out:     # the caller keeps this tensor: a fresh one (torch's caching allocator: no hipMalloc in steady state)
            return torch.empty((1, self.H, self.W, int(C)), dtype=dtype, device=self.ctx.device)
        return self.ctx.out(C, dtype)

    # the builders of EventBatch with the pooled output tensor as default destination
    def optimized(self, scale=1.0, dtype=torch.float64):
        return self.batch.optimized(scale=scale, dtype=dtype, out=self.out(12, dtype))

    def mdes(self, windows, funcs, aggs, scale=1.0, dtype=torch.float64, stacking="SBN"):
        C = len(windows)
        return self.batch.mdes(windows, funcs, aggs, scale, dtype, out=self.out(C, dtype) if C <= _lib.MAX_CHANNELS else None,
                               stacking=stacking)

    def event_stack(self, stack_size=12, premap=True, scale=1.0):
        return self.batch.event_stack(stack_size, premap, scale, out=self.out(stack_size, torch.float32))

    def time_surface(self, slices=6, tau=50000.0, premap=1, scale=1.0, dtype=torch.float64, indices=None):
        return self.batch.time_surface(slices, tau, premap, scale, dtype, out=self.out(2 * slices, dtype), indices=indices)

    def voxel(self, bins=5, mode=0, scale=1.0, t_range=None):
        return self.batch.voxel(bins, mode, scale, out=self.out(bins, torch.float64), t_range=t_range)

    def tore_full(self, k=6, frame_mode=0, scale=1.0):
        """TORE into the pooled (1, H, W, 2k) tensor WITHOUT the bounding-box read-back: finish(..., tore_k=k) cuts the
        frame out on the host from the statistics that travel with the result."""
        out = self.out(2 * k, torch.float32)
        b = self.batch
        b.bin()
        null = ctypes.c_void_p(None)
        with b._dev():
            _lib.check(b.lib.evrep_tore_ftime(*b._args(), int(k), int(frame_mode), null, null, null, float(scale), _ptr(out),
                                              b._sp()), "evrep_tore_ftime")
        return out


def sample_batch(event_sequence, height, width, truncate=False, rebase_t=False, device_out=False):
    """Structured x,y,t,p record array (or (n,4) array) -> the pooled one-window batch, events on their way to the GPU.
    device_out: the builders write into a FRESH device tensor and finish() returns it (no D2H)."""
    _require_gpu()                 # no CPU fallback: without the HIP library or a GPU the wrappers raise
    _lib.load()
    if isinstance(event_sequence, np.ndarray) and event_sequence.dtype.names:
        ev = from_structured(event_sequence, truncate=truncate, rebase_t=rebase_t)
    else:
        ev = np.ascontiguousarray(np.asarray(event_sequence), dtype=np.int32).reshape(-1, 4)
    n = int(ev.shape[0])
    ctx = _context(height, width, n)
    b = ctx.batch
    ctx.stream.synchronize()      # the previous call's H2D has left the staging buffer (its own finish() synchronised: free)
    ctx.ev_np[:n] = ev
    ctx.off_np[1] = n
    ctx.off_host_np[1] = n
    ctx.buf_dev[:n + 1].copy_(ctx.buf_pinned[:n + 1], non_blocking=True)     # offsets + events: one H2D
    b._binned = False
    return SampleBatch(ctx, n, device_out)


def finish(sb, dev_out, allow_oob=False, what="builder", out=None, tore_k=None, allow_unsorted=False):
    """Status + result in ONE synchronisation.  dev_out: the (1, H, W, C) device tensor a builder filled.  Returns the
    (H, W, C) numpy array (TORE: the bounding-box frame) -- the caller's own, see the module docstring -- or, for a
    ``device_out`` sample, the CUDA tensor itself; raises what the reference raises."""
    ctx, b = sb.ctx, sb.batch
    b.bin()
    with b._dev():
        _lib.check(b.lib.evrep_copy_window_meta_async(ctypes.byref(b.plan), _ptr(b.workspace), _ptr(ctx.meta_pinned),
                                                      b._sp()), "evrep_copy_window_meta_async")
    shape = tuple(dev_out.shape[1:])
    slot = host = None
    if not sb.device_out:
        slot = ctx.result_slot(shape, dev_out.dtype) if out is None else None
        host = slot.tensor if slot is not None else ctx.staging_buffer(shape, dev_out.dtype)
        host.copy_(dev_out[0], non_blocking=True)
    ctx.stream.synchronize()
    meta = ctx.meta_np
    _raise_for_status_word(int(meta[8]) & 0xffffffff, b, allow_oob, what, allow_unsorted)
    if sb.device_out:
        res = dev_out[0]
        if tore_k is not None:
            hb, wb = int(meta[5]) - int(meta[4]) + 1, int(meta[3]) - int(meta[2]) + 1
            res = res.reshape(-1)[: hb * wb * 2 * tore_k].view(hb, wb, 2 * tore_k)
        return res
    arr = slot.master.view() if slot is not None else host.numpy()      # a view whose base is the slot's master array
    if tore_k is not None:          # the events' bounding box, origin-shifted (gen1_transforms.py:61-64): a prefix of the buffer
        xmin, xmax, ymin, ymax = int(meta[2]), int(meta[3]), int(meta[4]), int(meta[5])
        hb, wb = ymax - ymin + 1, xmax - xmin + 1
        arr = arr.reshape(-1)[: hb * wb * 2 * tore_k].reshape(hb, wb, 2 * tore_k)
    if out is not None:
        np.copyto(out, arr)
        return out
    return arr if slot is not None else arr.copy()      # the pool is exhausted (the caller holds it all): a plain fresh array


def _raise_for_hot_overflow(st, what):
    if st & _lib.ST_HOT_OVERFLOW:   # cannot happen within evrep_plan_init's bounds; never hand out unwritten pixels
        raise _lib.EvrepError("%s: the workspace's hot-unit list overflowed (EVREP_ST_HOT_OVERFLOW): the tensor is incomplete" % what)


def _raise_for_status_word(st, batch, allow_oob, what, allow_unsorted=False):
    _raise_for_hot_overflow(st, what)
    if st & _lib.ST_EMPTY:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")  # t.min() on no events
    if (st & _lib.ST_OOB) and not allow_oob:
        raise IndexError("%s: event coordinates outside the %dx%d frame" % (what, batch.W, batch.H))
    if (st & _lib.ST_UNSORTED) and not allow_unsorted:
        # MixedDensityEventStack works in ARRAY order (windows by index, scatter in index order, t.min() / t.max()): the
        # kernels do the same whatever the timestamps' order, so its wrappers pass allow_unsorted; the other builders'
        # semantics on unsorted input differ from what the FIFO / last-event kernels compute
        raise UnsortedWindow("%s: timestamps must be ascending (the reference's adapters deliver them so)" % what)


def events_from_fields(x, y, t, p, truncate=False):
    """Four field arrays -> (n, 4) int32 rows; see synthetic.narrow_to_int32 for the float / range rules."""
    n = len(x)
    ev = np.empty((n, 4), dtype=np.int32)
    for k, (name, col) in enumerate(zip("xytp", (x, y, t, p))):
        ev[:, k] = narrow_to_int32(col, name, truncate)
    return ev


def single_batch(event_sequence, height, width, truncate=False, rebase_t=False):
    """Structured x,y,t,p record array (or (n,4) array) -> one-window EventBatch."""
    if isinstance(event_sequence, np.ndarray) and event_sequence.dtype.names:
        ev = from_structured(event_sequence, truncate=truncate, rebase_t=rebase_t)
    else:
        ev = np.ascontiguousarray(np.asarray(event_sequence), dtype=np.int32).reshape(-1, 4)
    return EventBatch.from_numpy(ev, height, width)


def raise_for_status(batch, allow_oob=False, what="builder", any_window_oob=False, allow_unsorted=False):
    """Turn the per-window status word into the exception the reference raises (window 0 is the sample;
    ``any_window_oob``: the out-of-frame check covers every window of the batch)."""
    sts = batch.status()
    st = int(sts[0])
    for s_ in sts:
        _raise_for_hot_overflow(int(s_), what)
    if st & _lib.ST_EMPTY:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")  # t.min() on no events
    oob = any(int(s) & _lib.ST_OOB for s in sts) if any_window_oob else (st & _lib.ST_OOB)
    if oob and not allow_oob:
        raise IndexError("%s: event coordinates outside the %dx%d frame" % (what, batch.W, batch.H))
    if (st & _lib.ST_UNSORTED) and not allow_unsorted:
        raise UnsortedWindow("%s: timestamps must be ascending (the reference's adapters deliver them so)" % what)
    return st
