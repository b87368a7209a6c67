"""Shared helpers of the per-sample wrappers (B = 1 batches on the current HIP device)."""
import numpy as np

from .. import _lib
from ..engine import EventBatch
from ..synthetic import from_structured, narrow_to_int32


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
