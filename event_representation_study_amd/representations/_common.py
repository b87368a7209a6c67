"""Shared helpers of the per-sample wrappers (B = 1 batches on the current HIP device).

r03: a per-sample call no longer allocates.  One pooled context per (process, device, H, W, size bucket) keeps the plan and
its workspace, a pinned staging buffer for the events, the device event buffer, the output tensors and a ring of pinned
result buffers; a call is  host convert -> async H2D -> bin -> build -> async copy of the window statistics -> async D2H of
the result -> ONE stream synchronisation -> the status word is checked and the exception the reference raises is raised.
The array a wrapper returns is a view of a pinned ring slot: it stays valid for the next RESULT_RING_DEPTH - 1 calls of the
same shape (the reference's callers resize / convert it at once); pass ``out=`` for a copy into your own array, or set
``EVREP_RESULT_RING=0`` for a fresh array per call (one more host copy of the result).
"""
import ctypes
import os

import numpy as np
import torch

from .. import _lib
from ..engine import EventBatch, _ptr, _require_gpu, _stream_ptr
from ..synthetic import from_structured, narrow_to_int32

RESULT_RING_DEPTH = int(os.environ.get("EVREP_RESULT_RING", "8"))
_CONTEXTS = {}


class _SampleContext:
    """Everything a B = 1 call needs, allocated once."""

    def __init__(self, height, width, cap, device):
        self.cap = int(cap)
        self.device = device
        self.ev_pinned = torch.empty((self.cap, 4), dtype=torch.int32, pin_memory=True)
        self.ev_dev = torch.empty((self.cap, 4), dtype=torch.int32, device=device)
        self.off_pinned = torch.zeros(2, dtype=torch.int64, pin_memory=True)
        self.batch = EventBatch(self.ev_dev, torch.tensor([0, self.cap], dtype=torch.int64), height, width,
                                max_events_per_window=self.cap)
        self.meta_pinned = torch.zeros(16, dtype=torch.int32, pin_memory=True)      # one 64-byte WindowMeta
        self.outs = {}          # (C, dtype) -> (1, H, W, C) device tensor
        self.rings = {}         # (shape, dtype) -> [pinned tensors], position

    def out(self, C, dtype):
        key = (int(C), dtype)
        if key not in self.outs:
            self.outs[key] = torch.empty((1, self.batch.H, self.batch.W, int(C)), dtype=dtype, device=self.device)
        return self.outs[key]

    def ring_slot(self, shape, dtype):
        key = (tuple(shape), dtype)
        ring = self.rings.setdefault(key, [[], -1])
        depth = max(RESULT_RING_DEPTH, 1)
        if len(ring[0]) < depth:
            ring[0].append(torch.empty(shape, dtype=dtype, pin_memory=True))
            return ring[0][-1]
        ring[1] = (ring[1] + 1) % depth
        return ring[0][ring[1]]


def _context(height, width, n):
    dev = torch.device("cuda", torch.cuda.current_device())
    cap = 4096
    while cap < n:
        cap *= 2
    key = (os.getpid(), str(dev), int(height), int(width), cap)      # per process: a HIP context does not survive fork()
    ctx = _CONTEXTS.get(key)
    if ctx is None:
        ctx = _CONTEXTS[key] = _SampleContext(height, width, cap, dev)
    return ctx


class SampleBatch:
    """One pooled window: ``.batch`` is the resident EventBatch (its builders take ``out=``), ``.out(C, dtype)`` the pooled
    output tensor, ``finish()`` the one synchronisation."""

    def __init__(self, ctx, n):
        self.ctx, self.batch, self.n = ctx, ctx.batch, int(n)
        self.H, self.W = ctx.batch.H, ctx.batch.W

    def out(self, C, dtype):
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

    def time_surface(self, slices=6, tau=50000.0, premap=True, scale=1.0, dtype=torch.float64, indices=None):
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
        with torch.cuda.device(b.device):
            _lib.check(b.lib.evrep_tore_ftime(*b._args(), int(k), int(frame_mode), null, null, null, float(scale), _ptr(out),
                                              _stream_ptr()), "evrep_tore_ftime")
        return out


def sample_batch(event_sequence, height, width, truncate=False, rebase_t=False):
    """Structured x,y,t,p record array (or (n,4) array) -> the pooled one-window batch, events on their way to the GPU."""
    _require_gpu()                 # no CPU fallback: without the HIP library or a GPU the wrappers raise
    _lib.load()
    if isinstance(event_sequence, np.ndarray) and event_sequence.dtype.names:
        ev = from_structured(event_sequence, truncate=truncate, rebase_t=rebase_t)
    else:
        ev = np.ascontiguousarray(np.asarray(event_sequence), dtype=np.int32).reshape(-1, 4)
    n = int(ev.shape[0])
    ctx = _context(height, width, n)
    b = ctx.batch
    stream = torch.cuda.current_stream(b.device)
    stream.synchronize()          # the previous call's H2D has left the staging buffer (its own finish() synchronised: free)
    ctx.ev_pinned[:n].numpy()[...] = ev
    ctx.off_pinned[1] = n
    ctx.ev_dev[:n].copy_(ctx.ev_pinned[:n], non_blocking=True)
    b.offsets.copy_(ctx.off_pinned, non_blocking=True)
    b.offsets_host = ctx.off_pinned.clone()
    b._binned = False
    return SampleBatch(ctx, n)


def finish(sb, dev_out, allow_oob=False, what="builder", out=None, tore_k=None, allow_unsorted=False):
    """Status + result in ONE synchronisation.  dev_out: the (1, H, W, C) device tensor a builder filled.  Returns the
    (H, W, C) numpy array (TORE: the bounding-box frame) or raises what the reference raises."""
    ctx, b = sb.ctx, sb.batch
    b.bin()
    with torch.cuda.device(b.device):
        _lib.check(b.lib.evrep_copy_window_meta_async(ctypes.byref(b.plan), _ptr(b.workspace), _ptr(ctx.meta_pinned),
                                                      _stream_ptr()), "evrep_copy_window_meta_async")
    host = ctx.ring_slot(tuple(dev_out.shape[1:]), dev_out.dtype)
    host.copy_(dev_out[0], non_blocking=True)
    torch.cuda.current_stream(b.device).synchronize()
    meta = ctx.meta_pinned.numpy()
    _raise_for_status_word(int(meta[8]) & 0xffffffff, b, allow_oob, what, allow_unsorted)
    arr = host.numpy()
    if tore_k is not None:          # the events' bounding box, origin-shifted (gen1_transforms.py:61-64): a prefix of the buffer
        xmin, xmax, ymin, ymax = int(meta[2]), int(meta[3]), int(meta[4]), int(meta[5])
        hb, wb = ymax - ymin + 1, xmax - xmin + 1
        arr = arr.reshape(-1)[: hb * wb * 2 * tore_k].reshape(hb, wb, 2 * tore_k)
    if out is not None:
        np.copyto(out, arr)
        return out
    return arr if RESULT_RING_DEPTH > 0 else arr.copy()


def _raise_for_status_word(st, batch, allow_oob, what, allow_unsorted=False):
    if st & _lib.ST_EMPTY:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")  # t.min() on no events
    if (st & _lib.ST_OOB) and not allow_oob:
        raise IndexError("%s: event coordinates outside the %dx%d frame" % (what, batch.W, batch.H))
    if (st & _lib.ST_UNSORTED) and not allow_unsorted:
        # MixedDensityEventStack works in ARRAY order (windows by index, scatter in index order, t.min() / t.max()): the
        # kernels do the same whatever the timestamps' order, so its wrappers pass allow_unsorted; the other builders'
        # semantics on unsorted input differ from what the FIFO / last-event kernels compute
        raise NotImplementedError("%s: timestamps must be ascending (the reference's adapters deliver them so)" % what)


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


def raise_for_status(batch, allow_oob=False, what="builder", any_window_oob=False):
    """Turn the per-window status word into the exception the reference raises (window 0 is the sample;
    ``any_window_oob``: the out-of-frame check covers every window of the batch)."""
    sts = batch.status()
    st = int(sts[0])
    if st & _lib.ST_EMPTY:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")  # t.min() on no events
    oob = any(int(s) & _lib.ST_OOB for s in sts) if any_window_oob else (st & _lib.ST_OOB)
    if oob and not allow_oob:
        raise IndexError("%s: event coordinates outside the %dx%d frame" % (what, batch.W, batch.H))
    if st & _lib.ST_UNSORTED:
        raise NotImplementedError("%s: timestamps must be ascending (the reference's adapters deliver them so)" % what)
    return st
