"""Seeded synthetic (x, y, t, p) event streams.

The shapes follow SURVEY.md section 8(d): ``x ~ U{0..W-1}``, ``y ~ U{0..H-1}``,
``t = sort(U{0..span-1})`` int32 microseconds (duplicates intended) and a
Bernoulli(1/2) polarity in {-1,+1} (or {0,1}).  Everything is drawn from
``numpy.random.default_rng(seed)`` so the golden fixtures, the tests and
``bench.py`` see identical streams.
"""
import numpy as np

EVENT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("t", "<i4"), ("p", "<i4")])


def make_events(n, width, height, seed=0, span_us=50000, polarity="pm1",
                dup_last=0, single_polarity=None):
    """Return an (n, 4) int32 array of [x, y, t, p] rows, time-sorted.

    polarity: "pm1" -> {-1,+1}; "01" -> {0,1}.
    dup_last: force the last ``dup_last`` events to share the final timestamp.
    single_polarity: if not None, every event gets this polarity value.
    """
    rng = np.random.default_rng(seed)
    ev = np.empty((n, 4), dtype=np.int32)
    ev[:, 0] = rng.integers(0, width, size=n)
    ev[:, 1] = rng.integers(0, height, size=n)
    t = np.sort(rng.integers(0, span_us, size=n)).astype(np.int32)
    if n:
        t -= t[0]                      # adapters hand over t[0] == 0 (gen1_2yolo.py:196)
    if dup_last and n:
        t[-dup_last:] = t[-1]
    ev[:, 2] = t
    pol = rng.integers(0, 2, size=n).astype(np.int32)
    if polarity == "pm1":
        pol = 2 * pol - 1
    elif polarity != "01":
        raise ValueError(polarity)
    if single_polarity is not None:
        pol[:] = single_polarity
    ev[:, 3] = pol
    return ev


def to_structured(ev):
    """(n,4) int32 -> the structured array the reference adapters hand to the
    builders (gen1_2yolo.py:567-571): fields x,y,t,p all '<i4'."""
    out = np.empty(ev.shape[0], dtype=EVENT_DTYPE)
    out["x"], out["y"], out["t"], out["p"] = ev[:, 0], ev[:, 1], ev[:, 2], ev[:, 3]
    return out


def from_structured(rec):
    """Structured x,y,t,p record array -> (n,4) int32 (values must be integral)."""
    n = rec.shape[0]
    ev = np.empty((n, 4), dtype=np.int32)
    for k, name in enumerate(("x", "y", "t", "p")):
        col = np.asarray(rec[name])
        if col.dtype.kind == "f":
            if n and not np.all(col == np.rint(col)):
                raise NotImplementedError(
                    "non-integral %r values are not supported by the int32 device layout" % name)
        ev[:, k] = col.astype(np.int64).astype(np.int32)
    return ev
