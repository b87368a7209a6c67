"""events_to_voxel_grid -- mirrors ev-licious/src/evlicious/tools/utils.py:52-108.

Only the numpy variant is a parity target (the reference's ``_cuda`` variant uses ``put_`` without
accumulate and is undefined for repeated coordinates).  Integer pixel coordinates only
(``Events.divider == 1``); the bilinear-in-x/y path for sub-pixel coordinates is not implemented.  The optional
``t0_us`` / ``t1_us`` range (:60-63) is honoured (``evrep_voxel_range``).
"""
import numpy as np
import torch

from .engine import EventBatch
from .synthetic import int64_to_int32


def events_to_voxel_grid(events, num_bins, normalize=True, t0_us=None, t1_us=None):
    """``events``: anything with ``x, y, t, p, width, height`` (ev-licious ``Events``; p in {-1,+1}).
    Returns a (num_bins, H, W) float32 grid: signed event counts per time bin (the reference computes
    its "bilinear" weight from the integer bin, utils.py:74, so the upper bin only receives zeros),
    optionally standardised over its non-zero entries."""
    H, W = int(events.height), int(events.width)
    n = len(events.x)
    grid = np.zeros((num_bins, H, W), np.float32)
    if n < 2:
        return grid
    x = np.asarray(events.x)
    if x.dtype.kind == "f":
        raise NotImplementedError("sub-pixel coordinates (divider > 1) are not supported")
    ev = np.empty((n, 4), np.int32)
    ev[:, 0], ev[:, 1] = x, np.asarray(events.y)
    t = np.asarray(events.t).astype(np.int64)
    base = int(t[0])
    ev[:, 2] = int64_to_int32(t - base, "t")          # the kernel only uses time differences
    ev[:, 3] = np.asarray(events.p)
    batch = EventBatch.from_numpy(ev, H, W)
    t_range = None
    if t0_us is not None or t1_us is not None:        # utils.py:60-63: the missing end defaults to t[0] / t[-1]
        t0 = int(t0_us) if t0_us is not None else int(t[0])
        t1 = int(t1_us) if t1_us is not None else int(t[-1])
        t_range = [[t0 - base, t1 - base]]
    if int(batch.status()[0]) & 2:
        raise AssertionError("event coordinates outside the sensor")     # Events.__init__ asserts this
    out = None
    for b0 in range(0, num_bins, 16):                  # 16 bins per launch would need bin offsets; keep it simple
        if num_bins > 16:
            raise NotImplementedError("num_bins > 16")
        out = batch.voxel(bins=num_bins, mode=2, t_range=t_range)[0]
    g = out.permute(2, 0, 1).to(torch.float32)
    if normalize:
        nz = g != 0
        if bool(nz.any()):
            vals = g[nz].to(torch.float64)
            mean, std = vals.mean(), vals.std(unbiased=False)
            if float(std) > 0:
                g[nz] = ((vals - mean) / (1e-5 + std)).to(torch.float32)
    return g.contiguous().cpu().numpy()
