"""ctypes front-end of oracle/evrep_oracle.c plus the numpy restatement of the GWD harness.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations are relative to /root/reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libevrep_oracle.so")

FUNCS = ["timestamp", "polarity", "count", "timestamp_pos", "timestamp_neg", "count_pos", "count_neg"]
AGGS = ["sum", "mean", "max", "variance"]

ERGO12 = (
    [0, 3, 2, 6, 5, 6, 2, 5, 1, 0, 4, 1],
    ["polarity", "timestamp_neg", "count_neg", "polarity", "count_pos", "count",
     "timestamp_pos", "count_neg", "timestamp_neg", "timestamp_pos", "timestamp", "count"],
    ["variance", "variance", "mean", "sum", "mean", "sum", "mean", "mean", "max", "max", "max", "mean"],
)


def build(force=False):
    src = os.path.join(_HERE, "evrep_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libevrep_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


class OracleIndexError(IndexError):
    pass


def _chk(rc):
    if rc == 1:
        raise OracleIndexError("index out of range (the reference raises here)")
    if rc != 0:
        raise ValueError("oracle: bad argument (rc=%d)" % rc)


def _ev(ev):
    ev = np.ascontiguousarray(ev, dtype=np.int32)
    assert ev.ndim == 2 and ev.shape[1] == 4
    return ev


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _ints(v):
    return (ctypes.c_int * len(v))(*v)


def mdes_windows(n):
    lo = (ctypes.c_int64 * 7)()
    hi = (ctypes.c_int64 * 7)()
    lib().oracle_mdes_windows(ctypes.c_int64(n), lo, hi)
    return list(lo), list(hi)


def mdes(ev, H, W, windows, funcs, aggs):
    """MixedDensityEventStack(...).stack -> (H, W, C) float64.  ``None`` entries -> zero channel."""
    ev = _ev(ev)
    C = len(windows)
    w = [-1 if v is None else int(v) for v in windows]
    f = [-1 if v is None else FUNCS.index(v) for v in funcs]
    a = [-1 if v is None else AGGS.index(v) for v in aggs]
    out = np.empty((H, W, C), dtype=np.float64)
    _chk(lib().oracle_mdes(_p(ev), ctypes.c_int64(ev.shape[0]), H, W, C, _ints(w), _ints(f), _ints(a), _p(out)))
    return out


def ergo12(ev, H, W):
    """get_optimized_representation -> (H, W, 12) float64."""
    ev = _ev(ev)
    out = np.empty((H, W, 12), dtype=np.float64)
    _chk(lib().oracle_ergo12(_p(ev), ctypes.c_int64(ev.shape[0]), H, W, _p(out)))
    return out


def event_stack(ev, H, W, stack_size=12, premap=True):
    ev = _ev(ev)
    out = np.empty((H, W, stack_size), dtype=np.float32)
    _chk(lib().oracle_event_stack(_p(ev), ctypes.c_int64(ev.shape[0]), H, W, stack_size, int(premap), _p(out)))
    return out


def time_surface(ev, H, W, slices=6, tau=50000.0, premap=True, return_idx=False):
    ev = _ev(ev)
    out = np.empty((H, W, 2 * slices), dtype=np.float64)
    idx = np.zeros(slices, dtype=np.int64)
    _chk(lib().oracle_time_surface(_p(ev), ctypes.c_int64(ev.shape[0]), H, W, slices,
                                   ctypes.c_double(tau), int(premap), _p(idx), _p(out)))
    return (out, idx) if return_idx else out


def tore_bbox(ev, k=6):
    """TORE as gen1_transforms.py:51-66 drives it: (Hbb, Wbb, 2k) float32."""
    ev = _ev(ev)
    hf, wf = ctypes.c_int(), ctypes.c_int()
    _chk(lib().oracle_tore_bbox(_p(ev), ctypes.c_int64(ev.shape[0]), k, ctypes.byref(hf), ctypes.byref(wf), None))
    out = np.empty((hf.value, wf.value, 2 * k), dtype=np.float32)
    _chk(lib().oracle_tore_bbox(_p(ev), ctypes.c_int64(ev.shape[0]), k, ctypes.byref(hf), ctypes.byref(wf), _p(out)))
    return out


def tore(x, y, ts, pol, sample_time, k, frame):
    """events2ToreFeature(x, y, ts, pol, sampleTimes, k, frameSize) with 1-based x, y."""
    x, y, ts, pol = (np.ascontiguousarray(v, dtype=np.int32) for v in (x, y, ts, pol))
    out = np.empty((frame[0], frame[1], 2 * k), dtype=np.float32)
    _chk(lib().oracle_tore(_p(x), _p(y), _p(ts), _p(pol), ctypes.c_int64(x.shape[0]),
                           ctypes.c_double(float(sample_time)), k, int(frame[0]), int(frame[1]), _p(out)))
    return out


def voxel(ev, H, W, bins=5):
    ev = _ev(ev)
    out = np.empty((H, W, bins), dtype=np.float64)
    _chk(lib().oracle_voxel(_p(ev), ctypes.c_int64(ev.shape[0]), H, W, bins, _p(out)))
    return out


def evl_voxel(ev, H, W, bins):
    """ev-licious events_to_voxel_grid(events, bins, normalize=False): (bins, H, W) float32."""
    ev = _ev(ev)
    out = np.empty((bins, H, W), dtype=np.float32)
    _chk(lib().oracle_evl_voxel(_p(ev), ctypes.c_int64(ev.shape[0]), H, W, bins, _p(out)))
    return out


def gwd(Xs, Xt, h=0.7):
    """OTMI(Xs, Xt, h).solve()[1] in closed form (PARITY UNPINNED for POT's part)."""
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    Xt = np.ascontiguousarray(Xt, dtype=np.float64)
    c = ctypes.c_double()
    _chk(lib().oracle_gwd(_p(Xs), ctypes.c_int64(Xs.shape[0]), Xs.shape[1], _p(Xt),
                          ctypes.c_int64(Xt.shape[0]), Xt.shape[1], ctypes.c_double(h), ctypes.byref(c)))
    return c.value


def otmi_point_clouds(events, rep, height, width, rep_size):
    """The quadrant harness of compute_otmi.py:96-205, numpy restatement: returns the list of
    (Xs float32 (n,4), Xt float64 (m,C+2)) pairs the reference hands to OTMI (3 of 4 quadrants)."""
    ev = np.array(events, dtype=np.int64)
    hx, hy = width / 2 - 1, height / 2 - 1
    x, y = ev[:, 0], ev[:, 1]
    quads = [
        ev[(x >= 0) & (x <= hx) & (y >= 0) & (y <= hy)],
        ev[(x > hx) & (x <= width - 1) & (y >= 0) & (y <= hy)],
        ev[(x >= 0) & (x <= hx) & (y > hy) & (y <= height - 1)],
        ev[(x > hx) & (x <= width - 1) & (y > hy) & (y <= height - 1)],
    ]
    sizes = [q.shape[0] for q in quads]
    skip = sizes.index(max(sizes))
    for q in quads[1:]:
        q[:, 0] -= q[:, 0].min()
        q[:, 1] -= q[:, 1].min()
    r2 = rep_size / 2 - 1
    xys = [([0, rep_size // 2 - 1], [0, r2]), ([r2, rep_size - 1], [0, r2]),
           ([0, r2], [r2, rep_size - 1]), ([r2, rep_size - 1], [r2, rep_size - 1])]
    pairs = []
    for i, q in enumerate(quads):
        if i == skip:
            continue
        # torch int64 / python int -> float32 (compute_otmi.py:164-169)
        xs = (q[:, 0].astype(np.float32) / np.float32((width - 1) // 2))
        ys = (q[:, 1].astype(np.float32) / np.float32((height - 1) // 2))
        t = q[:, 2]
        t = (t - t[0]).astype(np.float32) / np.float32(t[-1] - t[0])
        p = q[:, 3]
        p = (p - p.min()).astype(np.float32) / np.float32(p.max() - p.min())
        mask = (q[:, 0] < (width - 1) // 2) & (q[:, 1] < (height - 1) // 2)
        Xs = np.stack([xs[mask], ys[mask], t[mask], p[mask]], axis=-1)
        cx, cy = xys[i]
        r = rep[int(cy[0]): int(cy[1]) + 1, int(cx[0]): int(cx[1]) + 1, :]
        xe = np.repeat(np.arange(r.shape[0]).reshape(-1, 1), r.shape[1], axis=1) / (r.shape[0] - 1)
        ye = np.repeat(np.arange(r.shape[1]).reshape(1, -1), r.shape[0], axis=0) / (r.shape[1] - 1)
        r = np.concatenate((r, xe[..., None], ye[..., None]), axis=2).reshape(-1, rep.shape[2] + 2)
        r = r[np.abs(r[:, :-2]).sum(-1) > 0]
        pairs.append((Xs, r.astype(np.float64)))
    return pairs


def otmi(events, rep, height, width, rep_size, h=0.7):
    return float(np.mean([gwd(a, b, h) for a, b in otmi_point_clouds(events, rep, height, width, rep_size)]))
