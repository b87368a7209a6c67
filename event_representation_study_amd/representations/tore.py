"""events2ToreFeature -- mirrors representations/tore.py:6-83 of the reference."""
import numpy as np
import torch

from ._common import raise_for_status
from ..engine import EventBatch
from ..synthetic import field_to_int64, int64_to_int32


def events2ToreFeature(x, y, ts, pol, sampleTimes, k, frameSize):
    """TORE volume for one sample time.  x, y are 1-based (the reference indexes ``[i - 1, j - 1]``);
    returns (frameSize[0], frameSize[1], 2k) float32: the k most recent log-intervals per polarity.  Timestamps need not
    be ascending: the kernel then keeps, per event in array order, what the reference's ``np.partition`` on its k-vector
    keeps (tore.py:22-25; the order of the kept values is the sorting numpy's -- see k_tore in csrc/evrep_builders.hip).

    Float inputs behave as in the reference: float coordinates cannot index (IndexError), so it falls into its
    ``except`` branch and truncates them with int() (tore.py:29-33); float timestamps are used as they are --
    ``currentSampleTime - ts`` in float64 (:21) -- which is how n_imagenet calls it (seconds, imagenet.py:1093-1103)."""
    Hf, Wf = int(frameSize[0]), int(frameSize[1])
    x, y, ts, pol = np.asarray(x), np.asarray(y), np.asarray(ts), np.asarray(pol)
    xi = field_to_int64(x, "x", truncate=True)                     # int(i), int(j)
    yi = field_to_int64(y, "y", truncate=True)
    if len(xi) and (xi.min() < 1 or yi.min() < 1):
        # the reference indexes [i - 1, j - 1] (tore.py:25,41): a coordinate below 1 is a NEGATIVE numpy index and wraps to
        # the far side of the frame; beyond the frame numpy raises IndexError (twice: the except branch repeats the access)
        if xi.min() < 1 - Wf or yi.min() < 1 - Hf:
            raise IndexError("events2ToreFeature: index out of bounds for the %dx%d frame" % (Wf, Hf))
        xi = np.where(xi < 1, xi + Wf, xi)
        yi = np.where(yi < 1, yi + Hf, yi)
    # the sample time may arrive as a Python number, a numpy scalar, a 0-d array or a 0-d tensor: normalise first (a
    # non-integral one handed as a 0-d array must not be truncated by int())
    st = float(np.asarray(sampleTimes.cpu() if isinstance(sampleTimes, torch.Tensor) else sampleTimes).reshape(()))
    float_time = bool(ts.dtype.kind == "f" and len(ts) and not np.all(ts == np.rint(ts))) or st != np.rint(st)
    n = len(xi)
    ev = np.empty((n, 4), dtype=np.int32)
    ev[:, 0], ev[:, 1] = int64_to_int32(xi - 1, "x"), int64_to_int32(yi - 1, "y")
    # `pol > 0` is all the reference looks at (:19,34); fractional polarities keep their sign
    ev[:, 3] = np.where(pol > 0, 1, np.where(pol < 0, -1, 0))
    if n == 0:
        out = np.zeros((Hf, Wf, 2 * k), dtype=np.float32)
        out[...] = np.float32(np.log(np.float32(500e6) + 1) - np.log(151))
        return out
    if float_time:
        ev[:, 2] = np.arange(n, dtype=np.int32) if np.all(np.diff(ts) >= 0) else -np.arange(n, dtype=np.int32)  # order marker only
        batch = EventBatch.from_numpy(ev, Hf, Wf)
        raise_for_status(batch, what="events2ToreFeature", allow_unsorted=True)   # array order, as the reference (r05)
        tf = torch.from_numpy(np.ascontiguousarray(ts, dtype=np.float64)).to(batch.device)
        rep = batch.tore(k=int(k), frame_mode=2, scale=1.0, times_f64=tf, sample_times_f64=[st])
    else:
        ev[:, 2] = int64_to_int32(field_to_int64(ts, "t"), "t")
        batch = EventBatch.from_numpy(ev, Hf, Wf)
        raise_for_status(batch, what="events2ToreFeature", allow_unsorted=True)   # array order, as the reference (r05)
        rep = batch.tore(k=int(k), frame_mode=2, scale=1.0, sample_times=[int(st)])
    batch.check_built("events2ToreFeature")      # the hot-list flag is raised BY the builder: read it after the build
    return rep[0].cpu().numpy()
