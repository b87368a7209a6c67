"""Device-resident batch engine: the fast path of the package.

A :class:`EventBatch` holds B windows of events as one concatenated ``(total, 4)`` int32 tensor
plus ``B+1`` offsets in HBM, runs the (y,x) binning pass once, and then builds any number of
representations from the binned stream -- all on the caller's HIP stream, no host sync.

The per-sample functions that mirror the reference's Python names
(``representations/*.py``) are thin wrappers over this class with B = 1.
"""
import contextlib
import ctypes
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import Plan, check


def _require_gpu():
    if not torch.cuda.is_available():
        raise _lib.EvrepError("no HIP device visible: the builders run on an MI355X only (no CPU fallback)")


_NO_GUARD = contextlib.nullcontext()


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class EventBatch:
    """B event windows resident on one GPU.

    events  : (total, 4) int32 cuda tensor, rows [x, y, t, p], each window time-sorted
    offsets : (B+1,) int64 tensor (host or device); window b = rows [offsets[b], offsets[b+1])
    plan_flags : _lib.PLAN_* bits (None: from the EVREP_BIN_* environment switches of the tests and A/B tools)
    pacing  : store pacing of the wide float64 builders, see evrep_plan_set_pacing (None: EVREP_PACING or automatic)
    """

    def __init__(self, events, offsets, height, width, max_events_per_window=None, plan_flags=None, pacing=None):
        _require_gpu()
        self.lib = _lib.load()
        if events.dtype != torch.int32 or events.dim() != 2 or events.shape[1] != 4 or not events.is_cuda:
            raise ValueError("events must be a (total, 4) int32 CUDA tensor")
        self.events = events.contiguous()
        self.device = events.device
        off_host = offsets.detach().cpu().to(torch.int64) if isinstance(offsets, torch.Tensor) else \
            torch.as_tensor(np.asarray(offsets, dtype=np.int64))
        if max_events_per_window is None:
            max_events_per_window = int((off_host[1:] - off_host[:-1]).max().item()) if off_host.numel() > 1 else 0
        self.offsets_host = off_host
        self.offsets = off_host.to(self.device)
        self.B = int(off_host.numel() - 1)
        self.H, self.W = int(height), int(width)
        self.total = int(self.events.shape[0])
        self.plan = Plan()
        flags = _lib.plan_flags_from_env() if plan_flags is None else int(plan_flags)
        check(self.lib.evrep_plan_init_ex(ctypes.byref(self.plan), self.B, self.H, self.W, self.total,
                                          int(max_events_per_window), flags), "evrep_plan_init_ex")
        pacing = _lib.pacing_from_env() if pacing is None else int(pacing)
        if pacing is not None:
            check(self.lib.evrep_plan_set_pacing(ctypes.byref(self.plan), pacing), "evrep_plan_set_pacing")
        nbytes = int(self.lib.evrep_workspace_bytes(ctypes.byref(self.plan)))
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._binned = False
        # The per-sample wrappers (representations/_common.py) keep one pooled batch per (host thread, device, stream) and only use
        # it there: they pin the stream pointer and skip the device guard (torch.cuda.current_stream() + torch.cuda.device() cost
        # ~8 us per call of the ~100 us a sample takes).  None: look both up per call, as every other user must.
        self._pinned_stream = None

    def _sp(self):
        return self._pinned_stream if self._pinned_stream is not None else _stream_ptr()

    def _dev(self):
        return _NO_GUARD if self._pinned_stream is not None else torch.cuda.device(self.device)

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def from_numpy(cls, windows, height, width, device="cuda:0"):
        """windows: list of (n_b, 4) int32 arrays (or one array) -> device batch."""
        _require_gpu()
        if isinstance(windows, np.ndarray):
            windows = [windows]
        ws = [np.ascontiguousarray(w, dtype=np.int32).reshape(-1, 4) for w in windows]
        offs = np.zeros(len(ws) + 1, dtype=np.int64)
        np.cumsum([w.shape[0] for w in ws], out=offs[1:])
        cat = np.concatenate(ws, axis=0) if ws else np.zeros((0, 4), np.int32)
        ev = torch.from_numpy(cat).to(device)
        if ev.shape[0] == 0:
            ev = torch.zeros((0, 4), dtype=torch.int32, device=device)
        return cls(ev, torch.from_numpy(offs), height, width)

    def _args(self):
        return ctypes.byref(self.plan), _ptr(self.events), _ptr(self.offsets), _ptr(self.workspace)

    def bin(self):
        """Run the (y,x) binning pass (idempotent)."""
        if not self._binned:
            with self._dev():
                check(self.lib.evrep_bin_events(*self._args(), self._sp()), "evrep_bin_events")
            self._binned = True
        return self

    def rebin(self):
        self._binned = False
        return self.bin()

    # ------------------------------------------------------------------ read-backs (sync)
    def status(self):
        self.bin()
        st = np.zeros(self.B, dtype=np.uint32)
        with self._dev():
            check(self.lib.evrep_read_status(ctypes.byref(self.plan), _ptr(self.workspace),
                                             st.ctypes.data_as(ctypes.c_void_p), self._sp()), "evrep_read_status")
        return st

    def check_built(self, what="builder"):
        """After a builder call (one synchronisation): raise if a builder kernel set EVREP_ST_HOT_OVERFLOW -- the
        workspace's hot-unit list filled up and a unit was left unwritten.  The builders are asynchronous and the flag is
        raised BY the builder launch, so it can only be seen after it; the per-sample wrappers (`finish`) read it with the
        result, batched callers call this where they synchronise anyway.  Returns the status words."""
        from ._lib import EvrepError, ST_HOT_OVERFLOW
        st = self.status()
        if any(int(v) & ST_HOT_OVERFLOW for v in st):
            raise EvrepError("%s: the workspace's hot-unit list overflowed (EVREP_ST_HOT_OVERFLOW): the tensor is incomplete" % what)
        return st

    def bbox(self):
        self.bin()
        bb = np.zeros((self.B, 4), dtype=np.int32)
        with self._dev():
            check(self.lib.evrep_read_bbox(ctypes.byref(self.plan), _ptr(self.workspace),
                                           bb.ctypes.data_as(ctypes.c_void_p), self._sp()), "evrep_read_bbox")
        return bb

    # ------------------------------------------------------------------ builders
    def _out(self, out, C, dtype):
        shape = (self.B, self.H, self.W, C)
        if out is None:
            return torch.empty(shape, dtype=dtype, device=self.device)
        if tuple(out.shape) != shape or out.dtype != dtype or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous %s tensor of shape %r on %s" % (dtype, shape, self.device))
        return out

    @staticmethod
    def _dt(dtype):
        if dtype == torch.float64:
            return _lib.F64
        if dtype == torch.float32:
            return _lib.F32
        raise ValueError("dtype must be torch.float64 or torch.float32")

    def sbt_windows(self):
        """The eight "SBT" windows of every window of the batch as rank ranges + their flags (device tensors).  Formed per
        call: a pooled batch is refilled with other events between calls."""
        bounds = torch.empty((self.B, 8, 2), dtype=torch.int32, device=self.device)
        flags = torch.empty((self.B, 2), dtype=torch.int32, device=self.device)
        with self._dev():
            check(self.lib.evrep_mdes_sbt_windows(_ptr(self.events), _ptr(self.offsets), self.B, self.H, self.W,
                                                  _ptr(bounds), _ptr(flags), self._sp()), "evrep_mdes_sbt_windows")
        return bounds, flags

    def mdes(self, windows, funcs, aggs, scale=1.0, dtype=torch.float64, out=None, stacking="SBN"):
        """MixedDensityEventStack.stack for every window -> (B, H, W, C).  ``None`` entries give the
        reference's failed-channel zeros.  More than 16 channels are built 16 at a time.
        stacking "SBN": windows 0..6 cut by event count (the reference's choice); "SBT": windows 0..7 cut by time."""
        if stacking not in ("SBN", "SBT"):
            raise ValueError("stacking_type %r" % (stacking,))
        self.bin()
        C = len(windows)
        w = [-1 if v is None else int(v) for v in windows]
        f = [-1 if v is None else (_lib.FUNCS.index(v) if isinstance(v, str) else int(v)) for v in funcs]
        a = [-1 if v is None else (_lib.AGGS.index(v) if isinstance(v, str) else int(v)) for v in aggs]
        if C <= _lib.MAX_CHANNELS:
            out = self._out(out, C, dtype)
            null = ctypes.c_void_p(None)
            bounds, flags = self.sbt_windows() if stacking == "SBT" else (None, None)
            with self._dev():
                check(self.lib.evrep_mdes_ex(*self._args(), C, _lib.int32_array(w), _lib.int32_array(f),
                                             _lib.int32_array(a), float(scale), self._dt(dtype), _ptr(out),
                                             _ptr(bounds) if bounds is not None else null,
                                             _ptr(flags) if flags is not None else null, self._sp()), "evrep_mdes_ex")
            return out
        parts = [self.mdes(w[i:i + 16], f[i:i + 16], a[i:i + 16], scale, dtype, stacking=stacking) for i in range(0, C, 16)]
        res = torch.cat(parts, dim=3)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def optimized(self, scale=1.0, dtype=torch.float64, out=None):
        """get_optimized_representation (ERGO-12) for every window -> (B, H, W, 12)."""
        self.bin()
        out = self._out(out, 12, dtype)
        with self._dev():
            check(self.lib.evrep_optimized(*self._args(), float(scale), self._dt(dtype), _ptr(out), self._sp()),
                  "evrep_optimized")
        return out

    def event_stack(self, stack_size=12, premap=True, scale=1.0, out=None):
        """EventStack levels -> (B, H, W, stack_size) float32.  premap True/1: p -> (p + 1) // 2 first (the
        dispatcher, gen1_transforms.py:34); False/0: p is {0, 1}, the kernel forms 2p - 1; 2: the p column
        already holds the final int8 polarity value (EventStack.pre_stack forms it on the host)."""
        self.bin()
        out = self._out(out, stack_size, torch.float32)
        with self._dev():
            check(self.lib.evrep_event_stack(*self._args(), int(stack_size), int(premap), float(scale),
                                             _ptr(out), self._sp()), "evrep_event_stack")
        return out

    def _i32_dev(self, values, per_window):
        """None -> NULL; else a (B, per_window) int32 device tensor (a list is taken as one row per window)."""
        if values is None:
            return None, ctypes.c_void_p(None)
        t = torch.as_tensor(np.asarray(values, dtype=np.int32)).reshape(-1)
        if t.numel() == per_window and self.B > 1:
            t = t.repeat(self.B)
        if t.numel() != self.B * per_window:
            raise ValueError("expected %d x %d int32 values" % (self.B, per_window))
        t = t.to(self.device)
        return t, _ptr(t)

    def time_surface(self, slices=6, tau=50000.0, premap=True, scale=1.0, dtype=torch.float64, out=None, indices=None, times_f64=None):
        """ToTimesurface for every window -> (B, H, W, 2*slices), channel c = 2*s + p.  indices=None: the
        dispatcher's cuts searchsorted(t_norm, 1..slices); else explicit event indices per window.
        premap: True / 1 = p -> int8((p+1)/2) first; + 2 = timestamps not ascending (array-order scan, per-slice exponentials;
        needs explicit indices)."""
        self.bin()
        out = self._out(out, 2 * slices, dtype)
        keep, iptr = self._i32_dev(indices, slices)
        fptr = ctypes.c_void_p(None)
        if times_f64 is not None:      # float64 timestamps, one per event (the events' own t column then only orders them)
            if times_f64.dtype != torch.float64 or times_f64.device != self.device or times_f64.numel() != self.total \
                    or not times_f64.is_contiguous():
                raise ValueError("times_f64 must be a contiguous float64 tensor with one entry per event on %s" % self.device)
            if indices is None:
                raise ValueError("times_f64 needs explicit indices")
            fptr = _ptr(times_f64)
        with self._dev():
            check(self.lib.evrep_time_surface_ftime(*self._args(), int(slices), iptr, fptr, float(tau), int(premap),
                                              float(scale), self._dt(dtype), _ptr(out), self._sp()),
                  "evrep_time_surface")
        return out

    def tore(self, k=6, frame_mode=0, scale=1.0, out=None, sample_times=None, times_f64=None, sample_times_f64=None):
        """TORE.  frame_mode 0 (bounding box, the gen1/gen4 dispatcher's behaviour) returns a list of
        per-window (Hbb, Wbb, 2k) views (needs one host sync for the boxes); modes 1/2 return
        (B, H, W, 2k)."""
        self.bin()
        out = self._out(out, 2 * k, torch.float32)
        keep, tptr = self._i32_dev(sample_times, 1)
        fptr, sfptr, keep_f = ctypes.c_void_p(None), ctypes.c_void_p(None), None
        if sample_times_f64 is not None and times_f64 is None:
            # the C ABI rejects the pair (EINVAL); silently falling back to the int32 sample times would return a
            # plausible but different tensor
            raise ValueError("sample_times_f64 needs times_f64 (float64 event times)")
        if times_f64 is not None:      # float64 timestamps, one per event (the events' own t column is then not used)
            if times_f64.dtype != torch.float64 or times_f64.device != self.device or times_f64.numel() != self.total \
                    or not times_f64.is_contiguous():
                raise ValueError("times_f64 must be a contiguous float64 tensor with one entry per event on %s" % self.device)
            fptr = _ptr(times_f64)
            if sample_times_f64 is not None:
                keep_f = torch.as_tensor(np.asarray(sample_times_f64, dtype=np.float64)).reshape(-1).to(self.device)
                if keep_f.numel() != self.B:
                    raise ValueError("sample_times_f64 must hold one time per window")
                sfptr = _ptr(keep_f)
        with self._dev():
            check(self.lib.evrep_tore_ftime(*self._args(), int(k), int(frame_mode), tptr, fptr, sfptr, float(scale), _ptr(out),
                                            self._sp()), "evrep_tore_ftime")
        if frame_mode != 0:
            return out
        bb = self.bbox()
        views = []
        for b in range(self.B):
            hb, wb = int(bb[b, 3]) - int(bb[b, 1]) + 1, int(bb[b, 2]) - int(bb[b, 0]) + 1
            if hb <= 0 or wb <= 0:                      # an empty window has no bounding box: an empty frame
                hb = wb = 0
            flat = out[b].reshape(-1)
            views.append(flat[: hb * wb * 2 * k].view(hb, wb, 2 * k))
        return views

    def voxel(self, bins=5, mode=0, scale=1.0, out=None, t_range=None):
        """Voxel grids -> (B, H, W, bins) float64.  mode 0: compute_repr; 1: tonic ToVoxelGrid; 2: ev-licious
        events_to_voxel_grid (t_range: optional (B, 2) int64 [t0_us, t1_us] per window, mode 2 only)."""
        self.bin()
        out = self._out(out, bins, torch.float64)
        rptr = ctypes.c_void_p(None)
        if t_range is not None:
            tr = torch.as_tensor(np.asarray(t_range, dtype=np.int64)).reshape(-1)
            if tr.numel() != 2 * self.B:
                raise ValueError("t_range must hold (t0, t1) for each of the %d windows" % self.B)
            tr = tr.to(self.device)
            rptr = _ptr(tr)
        with self._dev():
            check(self.lib.evrep_voxel_range(*self._args(), int(bins), int(mode), float(scale), rptr, _ptr(out),
                                             self._sp()), "evrep_voxel_range")
        return out

    def voxel_tnorm(self, tnorm, bins=5, scale=1.0, out=None):
        """compute_repr with the caller's own normalised time (gromov_wasserstein.py:72-82) -> (B, H, W, bins) float64.
        tnorm: float64 device tensor, one value per event (indexed like the events)."""
        self.bin()
        if tnorm.dtype != torch.float64 or tnorm.device != self.device or tnorm.numel() != self.total or not tnorm.is_contiguous():
            raise ValueError("tnorm must be a contiguous float64 tensor with one entry per event on %s" % self.device)
        out = self._out(out, bins, torch.float64)
        with self._dev():
            check(self.lib.evrep_voxel_tnorm(*self._args(), _ptr(tnorm), int(bins), float(scale), _ptr(out), self._sp()),
                  "evrep_voxel_tnorm")
        return out

    def voxel_subpixel(self, xy, bins=5, t_range=None, out=None):
        """ev-licious events_to_voxel_grid for sub-pixel coordinates -> (B, H, W, bins) float32.  The batch holds the
        truncated coordinates; xy: (total, 2) float64 device tensor with every event's original (x, y)."""
        self.bin()
        if xy.dtype != torch.float64 or xy.device != self.device or tuple(xy.shape) != (self.total, 2) or not xy.is_contiguous():
            raise ValueError("xy must be a contiguous (total, 2) float64 tensor on %s" % self.device)
        out = self._out(out, bins, torch.float32)
        rptr, keep = ctypes.c_void_p(None), None
        if t_range is not None:
            keep = torch.as_tensor(np.asarray(t_range, dtype=np.int64)).reshape(-1).to(self.device)
            if keep.numel() != 2 * self.B:
                raise ValueError("t_range must hold (t0, t1) for each of the %d windows" % self.B)
            rptr = _ptr(keep)
        with self._dev():
            check(self.lib.evrep_voxel_subpixel(*self._args(), _ptr(xy), int(bins), rptr, _ptr(out), self._sp()),
                  "evrep_voxel_subpixel")
        return out

    def polstats(self, tnorm, pol, stat, tau=0.3, out=None):
        """n_imagenet's per-polarity accumulators (imagenet.py:169-511): channel c = stat[c] over the events of
        polarity class pol[c] at each pixel.  tnorm: float64 device tensor, one normalised time per event."""
        self.bin()
        pol = np.ascontiguousarray(pol, dtype=np.int32)
        stat = np.ascontiguousarray(stat, dtype=np.int32)
        if pol.shape != stat.shape or pol.ndim != 1:
            raise ValueError("pol and stat must be 1-D and of equal length")
        if tnorm.dtype != torch.float64 or tnorm.device != self.device or tnorm.numel() != self.total \
                or not tnorm.is_contiguous():
            raise ValueError("tnorm must be a contiguous float64 tensor with one entry per event on %s" % self.device)
        out = self._out(out, len(pol), torch.float32)
        with self._dev():
            check(self.lib.evrep_polstats(*self._args(), _ptr(tnorm), len(pol), pol.ctypes.data_as(ctypes.c_void_p),
                                          stat.ctypes.data_as(ctypes.c_void_p), float(tau), _ptr(out), self._sp()),
                  "evrep_polstats")
        return out

    def est_voxel(self, tnorm, C, segments, buckets, lo, hi, out=None):
        """EST quantisation layer forward (learned_repr.py:143-179) -> (B, H, W, 2C) float32; the value MLP is
        passed as its piecewise-linear table (est.PiecewiseLinearKernel.device_table)."""
        self.bin()
        if tnorm.dtype != torch.float32 or tnorm.device != self.device or tnorm.numel() != self.total \
                or not tnorm.is_contiguous():
            raise ValueError("tnorm must be a contiguous float32 tensor with one entry per event on %s" % self.device)
        if segments.dtype != torch.float64 or segments.dim() != 2 or segments.shape[1] != 3 or buckets.dtype != torch.int32:
            raise ValueError("segments must be float64 (nseg, 3), buckets int32")
        out = self._out(out, 2 * int(C), torch.float32)
        with self._dev():
            check(self.lib.evrep_est_voxel(*self._args(), _ptr(tnorm), int(C), _ptr(segments), int(segments.shape[0]),
                                           _ptr(buckets), int(buckets.numel()), float(lo), float(hi), _ptr(out),
                                           self._sp()), "evrep_est_voxel")
        return out


class BinBuildPipeline:
    """Throughput path for a STREAM of batches: the binning pass of batch k+1 runs on a side HIP stream
    while the builder of batch k runs on the caller's stream (the binning kernels are latency-bound and
    barely touch HBM, the builders are HBM-bound, so they overlap almost for free).  Each EventBatch
    owns its workspace, so two batches in flight = double buffering.

        pipe = BinBuildPipeline(device)
        for batch, out in work:              # e.g. alternating between two resident batches
            pipe.submit(batch, lambda b: b.optimized(out=out))
        pipe.drain()

    A batch's events may be refilled (on the caller's stream) once its build has been issued, i.e. after the
    submit() that followed its own; they must not change while its build is still pending.
    """

    def __init__(self, device=None):
        _require_gpu()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.bin_stream = torch.cuda.Stream(device=self.device)
        self._pending = None          # (batch, build_fn, binned_event)
        # batch -> event recorded after the last build that read its workspace; weak keys: a recycled id() of a
        # collected batch must not hand its event to an unrelated new batch
        self._built = weakref.WeakKeyDictionary()

    def _start_bin(self, batch):
        main = torch.cuda.current_stream(self.device)
        done = self._built.get(batch)
        with torch.cuda.stream(self.bin_stream):
            # the caller may have (re)filled batch.events on its own stream since the last build: ALWAYS order the
            # binning pass behind the caller's stream, and behind the last build that read this workspace
            self.bin_stream.wait_stream(main)
            if done is not None:
                self.bin_stream.wait_event(done)
            batch.rebin()
            ev = torch.cuda.Event()
            ev.record(self.bin_stream)
        return ev

    def submit(self, batch, build_fn):
        """Queue `batch`: its binning starts now on the side stream; the PREVIOUS submission is built now.
        Submitting the batch that is still pending builds it first (its workspace cannot be re-binned while a
        build of it is outstanding)."""
        if self._pending is not None and self._pending[0] is batch:
            self._flush()
        ev = self._start_bin(batch)
        self._flush()
        self._pending = (batch, build_fn, ev)

    def _flush(self):
        if self._pending is None:
            return None
        batch, build_fn, ev = self._pending
        self._pending = None
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        res = build_fn(batch)
        done = torch.cuda.Event()
        done.record(main)
        self._built[batch] = done
        return res

    def drain(self):
        return self._flush()


def probe_output_placement(shape, dtype, launch=None, candidates=8, launches=10, device="cuda:0", keep_first=False,
                           good_GBps=None, max_candidates=None):
    """Where a large output tensor lies in HBM decides up to 25 % of a builder launch on MI355X (NOTES.md 8: the same
    launch takes 134, 148 or 172 us into different 900 MiB allocations of one process; a linear fill does not care).
    A producer that allocates its output ring once can pick its allocations: this helper allocates `candidates` tensors,
    times a writer into each and returns (best tensor, its us per launch, all timings); the other tensors are released.
    The writer is `launch(out)` if given, else the library's placement probe (evrep_probe_store: the write footprint of
    the float64 12-channel builder and nothing else -- it overwrites the candidates with zeros).
    Fast regions come in runs and some processes find none among their first allocations: with good_GBps set, further
    rounds of `candidates` allocations follow (all kept alive meanwhile, so every round sees new regions) until one
    reaches that rate or max_candidates tensors have been tried.
    keep_first: also return the FIRST candidate (what a caller that does not probe would have got), as a 4th element."""
    _require_gpu()
    device = torch.device(device)
    lib = _lib.load()
    if launch is None:
        def launch(o):
            with torch.cuda.device(device):
                check(lib.evrep_probe_store(_ptr(o), o.numel() * o.element_size(), _stream_ptr()), "evrep_probe_store")
    outs, times = [], []
    limit = int(max_candidates) if max_candidates else int(candidates)
    while len(outs) < limit:
        for _ in range(min(int(candidates), limit - len(outs))):
            o = torch.empty(shape, dtype=dtype, device=device)
            for _ in range(3):
                launch(o)
            torch.cuda.synchronize(device)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(int(launches)):
                launch(o)
            b.record()
            torch.cuda.synchronize(device)
            outs.append(o)
            times.append(a.elapsed_time(b) / launches * 1e3)
        nbytes = outs[0].numel() * outs[0].element_size()
        if good_GBps is None or nbytes / (min(times) * 1e-6) / 1e9 >= good_GBps:
            break
    k = min(range(len(outs)), key=lambda i: times[i])
    best, first = outs[k], outs[0]
    del outs
    return (best, times[k], times, first) if keep_first else (best, times[k], times)


def gwd_padded_l1(Xs, Xt, h=0.7, out=None):
    """OTMI(Xs, Xt, h).solve()[1] on the GPU: Xs (n, ds), Xt (m, dt) array-likes -> 0-dim float64 cuda tensor.
    `out`: optional one-element float64 cuda tensor (e.g. ``costs[i:i+1]``) the kernel writes into directly --
    no host synchronisation, so a list of solves queues back to back."""
    _require_gpu()
    lib = _lib.load()
    dev = Xs.device if isinstance(Xs, torch.Tensor) and Xs.is_cuda else torch.device("cuda", torch.cuda.current_device())
    a = torch.as_tensor(np.asarray(Xs) if not isinstance(Xs, torch.Tensor) else Xs).to(dev, torch.float64).contiguous()
    b = torch.as_tensor(np.asarray(Xt) if not isinstance(Xt, torch.Tensor) else Xt).to(dev, torch.float64).contiguous()
    if a.dim() != 2 or b.dim() != 2 or a.shape[0] == 0 or b.shape[0] == 0:
        raise ValueError("Xs and Xt must be non-empty 2-D point clouds")
    n, m = int(a.shape[0]), int(b.shape[0])
    scratch = torch.empty(int(lib.evrep_gwd_scratch_bytes(n, m)), dtype=torch.uint8, device=dev)
    if out is not None:
        if out.dtype != torch.float64 or out.numel() != 1 or not out.is_cuda:
            raise ValueError("out must be a one-element float64 CUDA tensor")
        cost = out
    else:
        cost = torch.empty((), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.evrep_gwd_padded_l1(_ptr(a), n, int(a.shape[1]), _ptr(b), m, int(b.shape[1]), float(h),
                                      _ptr(scratch), _ptr(cost), _stream_ptr()), "evrep_gwd_padded_l1")
    return cost


class GwdWorkspace:
    """Scratch of the batched GWD solves and of the device harness, allocated once and grown on demand (a solve used to
    allocate its scratch per call and synchronise on its scalar)."""

    def __init__(self, device=None):
        _require_gpu()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._bufs = {}

    def get(self, name, nbytes):
        buf = self._bufs.get(name)
        if buf is None or buf.numel() < nbytes:
            buf = self._bufs[name] = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return buf

    def typed(self, name, shape, dtype):
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        return self.get(name, max(n, 256))[:n].view(dtype).view(*shape)


_WORKSPACES = {}


def _default_workspace(device):
    key = str(device)
    if key not in _WORKSPACES:
        _WORKSPACES[key] = GwdWorkspace(device)
    return _WORKSPACES[key]


def gwd_padded_l1_batch(Xs, n, Xt, m, n_cap, m_cap, xs_row=None, xt_row=None, h=0.7, out=None, workspace=None):
    """P solves of OTMI(Xs, Xt, h).solve()[1] in ONE call (five launches in all, no host synchronisation).
    Xs: (rows, ds) float64 cuda tensor holding every source cloud, pair p = rows [xs_row[p], xs_row[p] + n[p])
    (xs_row None: p * n_cap); Xt / xt_row / m likewise.  n, m, xs_row, xt_row: int64 cuda tensors of length P (the sizes
    may come straight from the device harness).  n_cap, m_cap: host upper bounds of the sizes.  Returns (P,) float64
    costs on the device (NaN for an empty or oversized cloud)."""
    _require_gpu()
    lib = _lib.load()
    dev = Xs.device
    P = int(n.numel())
    for t, nm in ((Xs, "Xs"), (Xt, "Xt")):
        if t.dtype != torch.float64 or t.dim() != 2 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("%s must be a contiguous 2-D float64 CUDA tensor" % nm)
    for t, nm in ((n, "n"), (m, "m"), (xs_row, "xs_row"), (xt_row, "xt_row")):
        if t is not None and (t.dtype != torch.int64 or t.numel() != P or not t.is_cuda or not t.is_contiguous()):
            raise ValueError("%s must be a contiguous int64 CUDA tensor of length %d" % (nm, P))
    ds, dt = int(Xs.shape[1]), int(Xt.shape[1])
    ws = workspace or _default_workspace(dev)
    nbytes = int(lib.evrep_gwd_batch_scratch_bytes(P, ds, dt, int(n_cap), int(m_cap)))
    if nbytes == 0:
        raise ValueError("bad batch geometry (P=%d, ds=%d, dt=%d, n_cap=%d, m_cap=%d)" % (P, ds, dt, n_cap, m_cap))
    scratch = ws.get("gwd_scratch", nbytes)
    if out is None:
        out = torch.empty(P, dtype=torch.float64, device=dev)
    elif out.dtype != torch.float64 or out.numel() != P or not out.is_cuda or not out.is_contiguous():
        raise ValueError("out must be a contiguous float64 CUDA tensor of length %d" % P)
    null = ctypes.c_void_p(None)
    with torch.cuda.device(dev):
        check(lib.evrep_gwd_padded_l1_batch(P, _ptr(Xs), _ptr(xs_row) if xs_row is not None else null, _ptr(n), ds,
                                            _ptr(Xt), _ptr(xt_row) if xt_row is not None else null, _ptr(m), dt,
                                            int(n_cap), int(m_cap), float(h), _ptr(scratch), _ptr(out), _stream_ptr()),
              "evrep_gwd_padded_l1_batch")
    return out


def otmi_event_clouds(events, offsets, height, width, cap=None, workspace=None):
    """Device half of otmi(): events (total, 4) int32 cuda tensor of B windows (offsets: (B+1,) int64, host or device) ->
    (Xs (B, 3, cap, 4) float64, n (B, 3) int64, quad (B, 3) int32), all on the device, nothing read back.
    The returned tensors are VIEWS of `workspace` (default: the device's shared GwdWorkspace): the next call with the same
    workspace overwrites them -- clone what must outlive it, or pass a GwdWorkspace of your own (one per host thread / stream)."""
    _require_gpu()
    lib = _lib.load()
    dev = events.device
    off_host = offsets.detach().cpu() if isinstance(offsets, torch.Tensor) else torch.as_tensor(np.asarray(offsets, dtype=np.int64))
    B = int(off_host.numel() - 1)
    if cap is None:
        cap = int((off_host[1:] - off_host[:-1]).max().item())
    off_dev = off_host.to(dev, torch.int64)
    ws = workspace or _default_workspace(dev)
    Xs = ws.typed("otmi_xs", (B, 3, int(cap), 4), torch.float64)
    n = ws.typed("otmi_n", (B, 3), torch.int64)
    quad = ws.typed("otmi_quad", (B, 3), torch.int32)
    scratch = ws.typed("otmi_ev_scratch", (int(lib.evrep_otmi_scratch_bytes(B)),), torch.uint8)
    with torch.cuda.device(dev):
        check(lib.evrep_otmi_event_clouds(_ptr(events), _ptr(off_dev), B, int(height), int(width), int(cap), _ptr(Xs),
                                          _ptr(n), _ptr(quad), _ptr(scratch), _stream_ptr()), "evrep_otmi_event_clouds")
    return Xs, n, quad


def otmi_rep_clouds(reps, quad, B, workspace=None, slot="otmi_xt"):
    """Device half of otmi(): reps (items, S, S, C) float64/float32 cuda tensor of letterboxed representations (item i
    belongs to window i % B), quad (B, 3) int32 from otmi_event_clouds -> (Xt (items, 3, m_cap, C + 2) float64,
    m (items, 3) int64).  Views of `workspace`, like otmi_event_clouds' results."""
    _require_gpu()
    lib = _lib.load()
    dev = reps.device
    reps = reps.contiguous()
    items, S, S2, C = (int(v) for v in reps.shape)
    if S != S2:
        raise ValueError("letterboxed representations are square")
    m_cap = (S - (S // 2 - 1)) ** 2
    ws = workspace or _default_workspace(dev)
    Xt = ws.typed(slot, (items, 3, m_cap, C + 2), torch.float64)
    m = ws.typed(slot + "_m", (items, 3), torch.int64)
    dt = {torch.float64: _lib.F64, torch.float32: _lib.F32}[reps.dtype]
    scratch = ws.typed("otmi_rep_scratch", (int(lib.evrep_otmi_scratch_bytes(items)),), torch.uint8)
    with torch.cuda.device(dev):
        check(lib.evrep_otmi_rep_clouds(_ptr(reps), dt, items, int(B), S, C, _ptr(quad), int(m_cap), _ptr(Xt), _ptr(m),
                                        _ptr(scratch), _stream_ptr()), "evrep_otmi_rep_clouds")
    return Xt, m, m_cap


def otmi_batch(events, offsets, reps, height, width, h=0.7, workspace=None):
    """otmi(events_b, rep_{r,b}, height, width, S) for R representations x B windows, entirely on the device:
    events (total, 4) int32 cuda tensor + offsets (B+1), reps (R, B, S, S, C) letterboxed representations ->
    (R, B) float64 cuda tensor of the mean cost over the three scored quadrants, plus the (R, B, 3) quadrant costs.
    The event clouds are built once per window and shared by the R representations."""
    R, B = int(reps.shape[0]), int(reps.shape[1])
    ws = workspace or _default_workspace(events.device)
    Xs, n, quad = otmi_event_clouds(events, offsets, height, width, workspace=ws)
    cap = int(Xs.shape[2])
    Xt, m, m_cap = otmi_rep_clouds(reps.reshape(R * B, *reps.shape[2:]), quad, B, workspace=ws)
    dev = events.device
    P = R * B * 3
    slot = torch.arange(B * 3, device=dev, dtype=torch.int64).repeat(R)       # pair p -> its window's (b, k) slot
    costs = gwd_padded_l1_batch(Xs.view(-1, 4), n.view(-1)[slot].contiguous(), Xt.view(-1, int(Xt.shape[-1])),
                                m.view(-1), cap, m_cap, xs_row=(slot * cap).contiguous(),
                                xt_row=(torch.arange(P, device=dev, dtype=torch.int64) * m_cap), h=h, workspace=ws)
    q = costs.view(R, B, 3)
    return q.mean(dim=2), q
