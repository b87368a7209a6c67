"""Shared helpers of the per-sample wrappers (B = 1 batches on the current HIP device)."""
import numpy as np

from .. import _lib
from ..engine import EventBatch
from ..synthetic import from_structured


def events_from_fields(x, y, t, p):
    n = len(x)
    ev = np.empty((n, 4), dtype=np.int32)
    for k, col in enumerate((x, y, t, p)):
        col = np.asarray(col)
        if col.dtype.kind == "f" and n and not np.all(col == np.rint(col)):
            raise NotImplementedError("non-integral event fields are not supported by the int32 device layout")
        ev[:, k] = col.astype(np.int64).astype(np.int32)
    return ev


def single_batch(event_sequence, height, width):
    """Structured x,y,t,p record array (or (n,4) array) -> one-window EventBatch."""
    if isinstance(event_sequence, np.ndarray) and event_sequence.dtype.names:
        ev = from_structured(event_sequence)
    else:
        ev = np.ascontiguousarray(np.asarray(event_sequence), dtype=np.int32).reshape(-1, 4)
    return EventBatch.from_numpy(ev, height, width)


def raise_for_status(batch, allow_oob=False, what="builder"):
    """Turn the per-window status word into the exception the reference raises."""
    st = int(batch.status()[0])
    if st & _lib.ST_EMPTY:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")  # t.min() on no events
    if (st & _lib.ST_OOB) and not allow_oob:
        raise IndexError("%s: event coordinates outside the %dx%d frame" % (what, batch.W, batch.H))
    if st & _lib.ST_UNSORTED:
        raise NotImplementedError("%s: timestamps must be ascending (the reference's adapters deliver them so)" % what)
    return st
