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


def mdes_sbt(ev, H, W, windows, funcs, aggs, return_masks=False):
    """MixedDensityEventStack(..., stacking_type="SBT").stack -> (H, W, C) float64: eight windows cut by normalised time."""
    ev = _ev(ev)
    C = len(windows)
    w = [-1 if v is None else int(v) for v in windows]
    f = [-1 if v is None else FUNCS.index(v) for v in funcs]
    a = [-1 if v is None else AGGS.index(v) for v in aggs]
    out = np.empty((H, W, C), dtype=np.float64)
    masks = np.empty((8, ev.shape[0]), dtype=np.uint8)
    _chk(lib().oracle_mdes_sbt(_p(ev), ctypes.c_int64(ev.shape[0]), H, W, C, _ints(w), _ints(f), _ints(a), _p(out), _p(masks)))
    return (out, masks) if return_masks else out


def ergo12(ev, H, W, out=None):
    """get_optimized_representation -> (H, W, 12) float64.  `out`: optional result buffer to reuse (the threaded
    CPU baseline of bench.py: a fresh 29.5 MB array per window makes the host's page-fault path the benchmark)."""
    ev = _ev(ev)
    if out is None:
        out = np.empty((H, W, 12), dtype=np.float64)
    _chk(lib().oracle_ergo12(_p(ev), ctypes.c_int64(ev.shape[0]), H, W, _p(out)))
    return out


def event_stack(ev, H, W, stack_size=12, premap=True):
    ev = _ev(ev)
    out = np.empty((H, W, stack_size), dtype=np.float32)
    _chk(lib().oracle_event_stack(_p(ev), ctypes.c_int64(ev.shape[0]), H, W, stack_size, int(premap), _p(out)))
    return out


def event_stack_split(x, y, p, t, last_timestamp, H, W, stack_size=12):
    """EventStack.pre_stack + post_stack for any last_timestamp (event_stack.py:15-68): the past half
    (t <= last) as is, the future half (t > last) reversed with negated polarity, each stacked on its own;
    the future half's level axis is reversed.  -> (H, W, 1 or 2, S) float32."""
    x = np.asarray(x).astype(np.int32)
    y = np.asarray(y).astype(np.int32)
    pv = 2 * np.asarray(p).astype(np.int8) - 1
    t = np.asarray(t).astype(np.int64)
    past, future = t <= last_timestamp, t > last_timestamp
    halves = [(x[past], y[past], pv[past])]
    if future.sum():
        halves.append((x[future][::-1], y[future][::-1], pv[future][::-1] * -1))
    levels = []
    for hx, hy, hp in halves:
        ev = np.zeros((len(hx), 4), np.int32)
        ev[:, 0], ev[:, 1], ev[:, 3] = hx, hy, hp
        levels.append(event_stack(ev, H, W, stack_size, premap=2))
    if len(levels) == 2:
        levels[1] = levels[1][:, :, ::-1]
    return np.stack(levels, axis=2)


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


def evl_voxel(ev, H, W, bins, t0_us=None, t1_us=None):
    """ev-licious events_to_voxel_grid(events, bins, normalize=False, t0_us, t1_us): (bins, H, W) float32."""
    ev = _ev(ev)
    out = np.empty((bins, H, W), dtype=np.float32)
    _chk(lib().oracle_evl_voxel_range(_p(ev), ctypes.c_int64(ev.shape[0]), H, W, bins,
                                      int(t0_us is not None), ctypes.c_int64(int(t0_us or 0)),
                                      int(t1_us is not None), ctypes.c_int64(int(t1_us or 0)), _p(out)))
    return out


def evl_voxel_subpixel(x, y, t, p, H, W, bins, t0_us=None, t1_us=None):
    """ev-licious events_to_voxel_grid(normalize=False) for non-uint16 (sub-pixel) coordinates
    (ev-licious/src/evlicious/tools/utils.py:52-108): per time-bin pass, per (xlim, ylim) tap, np.add.at into a
    float32 grid -- restated with explicit loops in the same order.  Small inputs only."""
    x, y = np.asarray(x), np.asarray(y)
    t, p = np.asarray(t).astype(np.int64), np.asarray(p)
    grid = np.zeros((bins, H, W), np.float32)
    if len(x) < 2:
        return grid
    t0 = t0_us if t0_us is not None else t[0]
    t1 = t1_us if t1_us is not None else t[-1]
    dT = t1 - t0
    if dT == 0:
        dT = 1.0
    tn = (bins - 1) * (t - t0) / dT
    ti = tn.astype("int32")
    xi, yi = x.astype("int32"), y.astype("int32")
    for dt_ in (0, 1):
        tl = ti + dt_
        wt = (1 - np.abs(tl - ti)) * p
        for dx in (0, 1):
            for dy in (0, 1):
                for k in range(len(x)):
                    if not (0 <= tl[k] < bins):
                        continue
                    X, Y = xi[k] + dx, yi[k] + dy
                    if X < 0 or Y < 0 or X >= W or Y >= H:
                        continue
                    w = (1 - np.abs(X - x[k])) * (1 - np.abs(Y - y[k]))
                    # np.add.at(float32 grid, ..., float64 values) runs the float64 add loop and rounds the sum back
                    grid[tl[k], Y, X] = np.float32(np.float64(grid[tl[k], Y, X]) + np.float64(w * wt[k]))
    return grid


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


# ---------------------------------------------------------------------------------------------
# F4: n_imagenet's per-polarity accumulators, numpy restatement
# (n_imagenet/real_cnn_model/data/imagenet.py; event_tensor = float64 (N,4) rows [x, y, t_seconds, p])
# ---------------------------------------------------------------------------------------------
NIMAGENET_EXP_TAU = 0.3  # imagenet.py:20


def _ni_split(ev, W):
    """pos / neg rows (:176-177), pixel index = x.long() + y.long()*W (:187,200), normalised time (:198-199)."""
    ev = np.asarray(ev, dtype=np.float64)
    pos, neg = ev[ev[:, 3] > 0], ev[ev[:, 3] < 0]
    idx = lambda a: a[:, 0].astype(np.int64) + a[:, 1].astype(np.int64) * W  # noqa: E731  (.long() truncates)
    start = ev[0, 2]
    length = ev[-1, 2] - ev[0, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        tp, tn = (pos[:, 2] - start) / length, (neg[:, 2] - start) / length
    return ev, idx(pos), idx(neg), tp, tn


def _ni_count(idx, H, W):
    return np.bincount(idx, minlength=H * W).reshape(H, W)  # torch.bincount (:187-189)


def _ni_scatter(src, idx, H, W, op):
    """torch_scatter.scatter_max / scatter_min with dim_size and no `out`: empty entries are 0."""
    o = np.full(H * W, -np.inf if op == "max" else np.inf)
    (np.maximum if op == "max" else np.minimum).at(o, idx, src)
    o[np.isinf(o)] = 0.0
    return o.reshape(H, W)


def nimagenet_acc(name, ev, H, W):
    """name in: acc, acc_time, acc_count, acc_count_pol, acc_count_only, acc_all, flat, flat_pol, acc_exp,
    acc_time_pol, acc_intensity -> (C, H, W) float32, as reshape_then_<name> (imagenet.py:169-511,841-871)."""
    ev = np.asarray(ev, dtype=np.float64)
    if len(ev) == 0:
        if name in ("acc_count", "acc_time_pol"):  # :258-261,483-486: ten synthetic events at the origin
            ev = np.zeros((10, 4))
            ev[:, 2] = (np.arange(10, dtype=np.float32) / np.float32(10.0)).astype(np.float64)
            ev[:, 3] = 1
        elif name == "acc_all":
            return np.zeros((6, 224, 224), np.float32)  # :353-354 (IMAGE_H, IMAGE_W, whatever H, W are)
        else:
            raise IndexError("empty event tensor")
    with np.errstate(divide="ignore", invalid="ignore"):
        if name in ("flat", "acc_count_only"):
            idx = ev[:, 0].astype(np.int64) + ev[:, 1].astype(np.int64) * W
            c = _ni_count(idx, H, W)
            out = [(c > 0).astype(np.float64)] if name == "flat" else [c]  # :403-406 / :334-336
            return np.stack(out).astype(np.float32)
        ev, ip, ineg, tp, tn = _ni_split(ev, W)
        pc, nc = _ni_count(ip, H, W), _ni_count(ineg, H, W)
        if name == "acc":  # :169-210
            pcn = pc.astype(np.float32) / np.float32(pc.max())  # int64 tensor / 0-dim float32 tensor -> float32
            ncn = nc.astype(np.float32) / np.float32(nc.max())
            ch = [pcn, _ni_scatter(tp, ip, H, W, "max"), ncn, _ni_scatter(tn, ineg, H, W, "max")]
        elif name == "acc_time":  # :213-247
            ch = [_ni_scatter(tp, ip, H, W, "min"), _ni_scatter(tp, ip, H, W, "max"),
                  _ni_scatter(tn, ineg, H, W, "min"), _ni_scatter(tn, ineg, H, W, "max")]
        elif name == "acc_count":  # :250-293
            ch = [pc, _ni_scatter(tp, ip, H, W, "max"), nc, _ni_scatter(tn, ineg, H, W, "max")]
        elif name == "acc_count_pol":  # :296-321
            ch = [pc, nc]
        elif name == "acc_all":  # :346-394
            ch = [pc, nc, _ni_scatter(tp, ip, H, W, "max"), _ni_scatter(tn, ineg, H, W, "max"),
                  _ni_scatter(tp, ip, H, W, "min"), _ni_scatter(tn, ineg, H, W, "min")]
        elif name == "flat_pol":  # :416-438
            ch = [(pc > 0).astype(np.float64), (nc > 0).astype(np.float64)]
        elif name == "acc_exp":  # :441-472
            ch = [np.exp(-(1 - _ni_scatter(tp, ip, H, W, "max")) / NIMAGENET_EXP_TAU),
                  np.exp(-(1 - _ni_scatter(tn, ineg, H, W, "max")) / NIMAGENET_EXP_TAU)]
        elif name == "acc_time_pol":  # :475-510
            ch = [_ni_scatter(tp, ip, H, W, "max"), _ni_scatter(tn, ineg, H, W, "max")]
        elif name == "acc_intensity":  # :841-870, float32 throughout
            it = pc.astype(np.float32) - nc.astype(np.float32)
            ch = [(it - it.min()) / (it.max() - it.min())]
        else:
            raise ValueError(name)
        return np.stack([np.asarray(c) for c in ch]).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# F4: EST quantisation layer forward, numpy restatement (ev-YOLOv6/yolov6/models/learned_repr.py:143-176)
# ---------------------------------------------------------------------------------------------
def est_voxel(events, dim, weights, slope=0.1):
    """events (N,5) [x,y,t,p,b] float32, weights = (w1,b1,W2,b2,w3,b3) -> (B, 2C, H, W) float32 before the
    letterbox: t /= t.max() per batch item (:159-160); for every bin i: values = t * mlp(t - i/(C-1)) (:167),
    accumulated at x + W*y + W*H*i + W*H*C*p + W*H*C*2*b in event order (:163-173).  The MLP runs in float32."""
    C, H, W = dim
    ev = np.asarray(events, dtype=np.float32)
    w1, b1, W2, b2, w3, b3 = [np.asarray(a, dtype=np.float32) for a in weights]
    x, y, p, b = (ev[:, k].astype(np.int64) for k in (0, 1, 3, 4))
    t = ev[:, 2].copy()
    nb = int(1 + ev[-1, 4])
    for bi in range(nb):
        m = b == bi
        if m.any():
            t[m] = t[m] / t[m].max()
    vox = np.zeros(2 * C * H * W * nb, dtype=np.float32)
    base = x + W * y + W * H * C * p + W * H * C * 2 * b
    leaky = lambda z: np.where(z > 0, z, np.float32(slope) * z)  # noqa: E731
    for i in range(C):
        u = t - np.float32(i / (C - 1))
        h1 = leaky(np.outer(u, w1.reshape(-1)) + b1.reshape(-1))
        h2 = leaky(h1 @ W2.T + b2.reshape(-1))
        f = h2 @ w3.reshape(-1) + np.float32(np.asarray(b3).reshape(-1)[0])
        np.add.at(vox, base + W * H * i, (t * f).astype(np.float32))
    vox = vox.reshape(nb, 2, C, H, W)
    return np.concatenate([vox[:, 0], vox[:, 1]], axis=1)
