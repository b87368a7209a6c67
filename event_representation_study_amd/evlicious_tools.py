"""events_to_voxel_grid -- mirrors ev-licious/src/evlicious/tools/utils.py:52-108.

Only the numpy variant is a parity target (the reference's ``_cuda`` variant uses ``put_`` without
accumulate and is undefined for repeated coordinates).  uint16 pixel coordinates
(``Events.divider == 1``) run the integer builder (``evrep_voxel`` mode 2); any other coordinate dtype -- sub-pixel
positions after ``resize_to_resolution`` -- takes the reference's bilinear-in-x/y draw (``evrep_voxel_subpixel``, float32
accumulation in the reference's tap order).  The optional ``t0_us`` / ``t1_us`` range (:60-63) is honoured.
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
    x, y = np.asarray(events.x), np.asarray(events.y)
    subpixel = x.dtype != np.uint16                    # utils.py:87-89: only uint16 coordinates take the integer path
    ev = np.empty((n, 4), np.int32)
    ev[:, 0], ev[:, 1] = x.astype("int32"), y.astype("int32")   # x_int, y_int (:92-93): truncation
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
    if num_bins > 16:
        raise NotImplementedError("num_bins > 16")
    if subpixel:
        # the bilinear draw into the four surrounding pixels (:95-98); integer-valued non-uint16 coordinates take
        # this path too, as in the reference (their second taps weigh exactly zero)
        xy = torch.from_numpy(np.stack([x.astype(np.float64), y.astype(np.float64)], axis=1)).to(batch.device)
        g = batch.voxel_subpixel(xy, bins=num_bins, t_range=t_range)[0].permute(2, 0, 1)
    else:
        g = batch.voxel(bins=num_bins, mode=2, t_range=t_range)[0].permute(2, 0, 1).to(torch.float32)
    if normalize:
        nz = g != 0
        if bool(nz.any()):
            vals = g[nz].to(torch.float64)
            mean, std = vals.mean(), vals.std(unbiased=False)
            if float(std) > 0:
                g[nz] = ((vals - mean) / (1e-5 + std)).to(torch.float32)
    return g.contiguous().cpu().numpy()
