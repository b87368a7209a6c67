"""compute_repr -- mirrors representation_search/gromov_wasserstein.py:72-82 (the in-repo voxel grid)."""
import numpy as np
import torch

from .._common import raise_for_status
from ...engine import EventBatch
from ...synthetic import field_to_int64, int64_to_int32


def compute_repr(x, y, t, p, width, height, bins=5):
    """The reference's own name and signature (gromov_wasserstein.py:72-82): ``x``, ``y`` integer pixel indices, ``t`` the
    CALLER's normalised time (the demo: ``(t - t[0]) / (t[-1] - t[0])``, :96 -- any float64 values are used as they are),
    ``p`` the polarity weights -> (height, width, bins) float64: ``b = (bins - 1) * t``; for ``blim`` in
    ``int(b), int(b) + 1``: ``grid[y, x, blim] += (1 - |blim - b|) * p`` where ``blim < bins`` -- the lower bin for every
    event, then the upper one, in array order (``np.add.at``).  Integral ``p`` only (the reference's callers hand 0 / 1)."""
    x, y, t, p = np.asarray(x), np.asarray(y), np.asarray(t, dtype=np.float64), np.asarray(p)
    n = len(x)
    if p.dtype.kind == "f" and n and not np.all(p == np.rint(p)):
        raise NotImplementedError("compute_repr: non-integral polarity weights")
    if n and ((t < 0).any() or not np.all(np.isfinite(t))):
        raise NotImplementedError("compute_repr: normalised times must be finite and >= 0 (a negative bin index wraps in numpy)")
    ev = np.empty((n, 4), dtype=np.int32)
    ev[:, 0], ev[:, 1] = int64_to_int32(field_to_int64(x, "x"), "x"), int64_to_int32(field_to_int64(y, "y"), "y")
    ev[:, 2] = np.arange(n, dtype=np.int32)          # array order is all the kernel needs of the events' own t column
    ev[:, 3] = int64_to_int32(field_to_int64(p, "p"), "p")
    if n == 0:
        return np.zeros((height, width, bins))
    batch = EventBatch.from_numpy(ev, height, width)
    raise_for_status(batch, what="compute_repr")
    tn = torch.from_numpy(np.ascontiguousarray(t)).to(batch.device)
    return batch.voxel_tnorm(tn, bins=bins)[0].cpu().numpy()


def compute_repr_from_events(events, width, height, bins=5):
    """Voxel grid of an (n, 4) int [x, y, t, p] window with the caller-side normalisation
    t = (t - t[0]) / (t[-1] - t[0]) of gromov_wasserstein.py:96 -> (H, W, bins) float64."""
    batch = EventBatch.from_numpy(np.asarray(events), height, width)
    raise_for_status(batch, what="compute_repr")
    return batch.voxel(bins=bins, mode=0)[0].cpu().numpy()
