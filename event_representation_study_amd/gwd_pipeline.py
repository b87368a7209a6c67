"""The caller side of the GWD score: representation -> keep-ratio resize ->
letterbox(114) -> ``otmi(events, rep)`` -> C_p, as ``Gen1H5GWD.__getitem__`` + ``measure_otmi`` do
(representations/representation_search/gen1_compute.py:30-104, ev-YOLOv6/yolov6/data/gen1_2yolo.py:230-265,
ev-YOLOv6/yolov6/data/data_augment.py:31-85).

The reference resizes with OpenCV (``cv2.resize`` INTER_AREA per channel when shrinking, INTER_LINEAR
otherwise).  OpenCV is absent here, so the two interpolations are RESTATED from OpenCV's published
algorithm (resize.cpp: ``computeResizeAreaTab`` / ``resizeArea_`` and the half-pixel-centre bilinear
table) as separable weight matrices applied on the GPU in float64 -- PARITY UNPINNED against cv2 itself
(different accumulation order; expected agreement ~1e-15 relative).
"""
import math

import numpy as np
import torch

from . import distributed
from .representations.representation_search.compute_otmi import otmi


def area_weights(src, dst):
    """(dst, src) float64 matrix of OpenCV's INTER_AREA weights for shrinking ``src`` samples to ``dst``."""
    scale = src / dst                                   # 1 / inv_scale, inv_scale = dsize / ssize
    Wm = np.zeros((dst, src), dtype=np.float64)
    for d in range(dst):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, src - f1)
        s1 = math.ceil(f1)
        s2 = min(math.floor(f2), src - 1)
        s1 = min(s1, s2)
        if s1 - f1 > 1e-3:
            Wm[d, s1 - 1] += (s1 - f1) / cell
        for sx in range(s1, s2):
            Wm[d, sx] += 1.0 / cell
        if f2 - s2 > 1e-3:
            Wm[d, s2] += min(min(f2 - s2, 1.0), cell) / cell
    return Wm


def linear_weights(src, dst):
    """(dst, src) float64 matrix of OpenCV's INTER_LINEAR weights (half-pixel centres, edge clamp)."""
    scale = src / dst
    Wm = np.zeros((dst, src), dtype=np.float64)
    for d in range(dst):
        fx = (d + 0.5) * scale - 0.5
        sx = math.floor(fx)
        fx -= sx
        if sx < 0:
            sx, fx = 0, 0.0
        if sx >= src - 1:
            sx, fx = src - 1, 0.0
        Wm[d, sx] += 1.0 - fx
        if fx:
            Wm[d, sx + 1] += fx
    return Wm


_TAPS = {}


def resize_taps(src, dst, interpolation, device):
    """(start, count, weights, T) device tables of the non-zero entries of the (dst, src) weight matrix: the few
    source samples every output sample integrates (2-3 for an area shrink by 1-2x, 2 for bilinear)."""
    key = (int(src), int(dst), interpolation, str(device))
    if key not in _TAPS:
        Wm = (area_weights if interpolation == "area" else linear_weights)(int(src), int(dst))
        nz = Wm != 0
        start = np.where(nz.any(1), nz.argmax(1), 0).astype(np.int32)
        last = np.where(nz.any(1), Wm.shape[1] - 1 - nz[:, ::-1].argmax(1), 0)
        count = np.where(nz.any(1), last - start + 1, 0).astype(np.int32)
        T = max(int(count.max()), 1)
        wt = np.zeros((int(dst), T), dtype=np.float64)
        for d in range(int(dst)):
            wt[d, :count[d]] = Wm[d, start[d]:start[d] + count[d]]
        _TAPS[key] = (torch.from_numpy(start).to(device), torch.from_numpy(count).to(device),
                      torch.from_numpy(wt).to(device), T)
    return _TAPS[key]


def resize_batch(rep, new_h, new_w, interpolation="area", scale=1.0, out_dtype=torch.float64, out=None):
    """rep: (B, H, W, C) float64/float32 CUDA tensor -> (B, new_h, new_w, C) of out_dtype; every channel on its
    own, like the reference's per-channel cv2.resize.  One launch of k_resize_taps (evrep_resize_taps)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    rep = rep.contiguous()
    B, H, W, C = (int(v) for v in rep.shape)
    ys, yc, yw, Ty = resize_taps(H, new_h, interpolation, rep.device)
    xs, xc, xw, Tx = resize_taps(W, new_w, interpolation, rep.device)
    T = max(Ty, Tx)
    if Ty != T:
        yw = torch.nn.functional.pad(yw, (0, T - Ty)).contiguous()
    if Tx != T:
        xw = torch.nn.functional.pad(xw, (0, T - Tx)).contiguous()
    if out is None:
        out = torch.empty((B, int(new_h), int(new_w), C), dtype=out_dtype, device=rep.device)
    dt = {torch.float64: _lib.F64, torch.float32: _lib.F32}
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(rep.device):
        _lib.check(lib.evrep_resize_taps(ptr(rep), dt[rep.dtype], B, H, W, C, int(new_h), int(new_w), T, ptr(ys), ptr(yc),
                                         ptr(yw), ptr(xs), ptr(xc), ptr(xw), float(scale), dt[out.dtype], ptr(out),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "evrep_resize_taps")
    return out


def resize(im, new_w, new_h, interpolation="area"):
    """im: (H, W, C) tensor/array -> (new_h, new_w, C) float64 cuda tensor; every channel on its own, like
    the reference's per-channel cv2.resize."""
    t = torch.as_tensor(im).to("cuda", torch.float64)
    return resize_batch(t[None], new_h, new_w, interpolation)[0]


def resize_image(im, img_size, augment=False):
    """Gen1H5.resize_image (gen1_2yolo.py:230-265): keep-ratio resize so the long side is ``img_size``."""
    h0, w0 = int(im.shape[0]), int(im.shape[1])
    r = img_size / max(h0, w0)
    if r == 1:
        return torch.as_tensor(im).to("cuda", torch.float64)
    interp = "area" if (r < 1 and not augment) else "linear"
    return resize(im, int(w0 * r), int(h0 * r), interp)


def letterbox(im, new_shape, color=114.0, scaleup=False):
    """letterbox(..., auto=False) of data_augment.py:31-85: pad (and, if needed, bilinearly resize) to a
    ``new_shape`` square with ``color``."""
    h, w = int(im.shape[0]), int(im.shape[1])
    r = min(new_shape / h, new_shape / w)
    if not scaleup:
        r = min(r, 1.0)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = (new_shape - nw) / 2, (new_shape - nh) / 2
    if (w, h) != (nw, nh):
        im = resize(im, nw, nh, "linear")
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    t = torch.as_tensor(im).to("cuda", torch.float64)
    out = torch.full((nh + top + bottom, nw + left + right, t.shape[2]), float(color), dtype=torch.float64, device=t.device)
    out[top:top + nh, left:left + nw] = t
    return out


def rep_for_gwd(rep, img_size):
    """Representation (H, W, C) -> the letterboxed (img_size, img_size, C) array ``otmi`` is given."""
    return letterbox(resize_image(rep, img_size), img_size)


def measure_cp(windows, build_rep, height, width, img_size=240):
    """C_p of one representation over a list of raw (n, 4) int windows: per window
    ``otmi(events, letterboxed rep)`` (mean of 3 quadrant solves), then the mean over windows
    (gen1_compute.py:91-104).  Windows are dealt to the ranks of the current process group and the
    per-window scalars are assembled with ONE all_gather."""
    def score(ev):
        rep = build_rep(ev)                                        # (H, W, C) tensor on the GPU
        lb = rep_for_gwd(rep, img_size).cpu().numpy()
        return float(otmi(torch.from_numpy(np.asarray(ev)), lb, height, width, img_size))

    scores = distributed.sharded_scores(score, list(windows))
    return float(scores.mean().item()), scores


# ------------------------------------------------------------------------------------------------ device path (r03)
def rep_for_gwd_batch(rep, img_size):
    """(B, H, W, C) representations on the GPU -> (B, img_size, img_size, C) float64, letterboxed: the batched form of
    rep_for_gwd (keep-ratio resize, INTER_AREA when shrinking, then letterbox(114) without scale-up), one resize launch."""
    rep = rep if rep.dtype == torch.float64 else rep.to(torch.float64)
    B, h0, w0, C = (int(v) for v in rep.shape)
    r = img_size / max(h0, w0)
    if r != 1:
        rep = resize_batch(rep, int(h0 * r), int(w0 * r), "area" if r < 1 else "linear")
    h, w = int(rep.shape[1]), int(rep.shape[2])
    r2 = min(min(img_size / h, img_size / w), 1.0)
    nw, nh = int(round(w * r2)), int(round(h * r2))
    if (w, h) != (nw, nh):
        rep = resize_batch(rep, nh, nw, "linear")
    dw, dh = (img_size - nw) / 2, (img_size - nh) / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = torch.full((B, nh + top + bottom, nw + left + right, C), 114.0, dtype=torch.float64, device=rep.device)
    out[:, top:top + nh, left:left + nw] = rep
    return out


def measure_cp_device(windows, builders, height, width, img_size=240, h=0.7, workspace=None):
    """C_p of several representations over the same windows, entirely on the device (BASELINE config 4): this rank's
    windows are ONE EventBatch; every builder is one launch over the batch; resize + letterbox are batched; the quadrant
    point clouds are built by the device harness (the event side once, shared by all representations); the 3 solves of
    every (representation, window) are scored by ONE batched GWD call per representation; the scores of all ranks meet
    in ONE all_gather.  builders: {name: fn(EventBatch) -> (B, H, W, C) tensor, or a list of B (Hb, Wb, C) tensors}.
    Returns {name: (C_p, per-window scores)}."""
    from . import engine as eng
    windows = list(windows)
    mine = distributed.shard_indices(len(windows))
    dev = torch.device("cuda", torch.cuda.current_device())
    out = {}
    if mine:
        wins = [np.ascontiguousarray(windows[i], dtype=np.int32).reshape(-1, 4) for i in mine]
        batch = eng.EventBatch.from_numpy(wins, height, width, device=dev)
        ws = workspace or eng._default_workspace(dev)
        Xs, n, quad = eng.otmi_event_clouds(batch.events, batch.offsets_host, height, width, workspace=ws)
        cap, B = int(Xs.shape[2]), len(wins)
    for name, build in builders.items():
        if mine:
            rep = build(batch)
            if isinstance(rep, (list, tuple)):                       # TORE: its own bounding-box frame per window
                lb = torch.cat([rep_for_gwd_batch(r[None], img_size) for r in rep])
            else:
                lb = rep_for_gwd_batch(rep, img_size)
            Xt, m, m_cap = eng.otmi_rep_clouds(lb, quad, B, workspace=ws)
            P = 3 * B
            costs = eng.gwd_padded_l1_batch(Xs.view(-1, 4), n.view(-1), Xt.view(-1, int(Xt.shape[-1])), m.view(-1), cap, m_cap,
                                            xs_row=torch.arange(P, device=dev, dtype=torch.int64) * cap,
                                            xt_row=torch.arange(P, device=dev, dtype=torch.int64) * m_cap, h=h, workspace=ws)
            local = costs.view(B, 3).mean(dim=1)
        else:
            local = torch.zeros(0, dtype=torch.float64, device=dev)
        scores = distributed.gather_vector(local, mine, len(windows))
        out[name] = (scores.mean(), scores)
    return {k: (float(v[0].item()), v[1]) for k, v in out.items()}   # the only host synchronisation
