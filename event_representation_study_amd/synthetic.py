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


I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1


def field_to_int64(col, name, truncate=False):
    """One event field -> int64.  Integer fields pass through; float fields must hold integral values unless
    ``truncate`` -- the reference's own ``.astype(np.int32)`` / ``.astype(np.int64)`` of
    MixedDensityEventStack.stack and EventStack.pre_stack (mixed_density_event_stack.py:26-29,
    event_stack.py:16-19), i.e. C truncation toward zero."""
    col = np.asarray(col)
    if col.dtype.kind == "f":
        if col.size and not np.all(np.isfinite(col)):
            raise OverflowError("non-finite %r values" % name)
        if col.size and not truncate and not np.all(col == np.rint(col)):
            raise NotImplementedError("non-integral %r values are not supported by the int32 device layout" % name)
        return np.trunc(col).astype(np.int64)
    if col.dtype.kind not in "iub":
        raise TypeError("event field %r has dtype %s" % (name, col.dtype))
    if col.dtype == np.uint64 and col.size and col.max() > np.iinfo(np.int64).max:
        raise OverflowError("event field %r exceeds int64" % name)
    return col.astype(np.int64)


def int64_to_int32(v, name):
    """Range-checked narrowing: values outside int32 raise OverflowError instead of wrapping silently."""
    if v.size and (v.min() < I32_MIN or v.max() > I32_MAX):
        raise OverflowError("event field %r exceeds the int32 range of the device layout "
                            "(absolute timestamps: subtract the window's first timestamp)" % name)
    return v.astype(np.int32)


def narrow_to_int32(col, name, truncate=False):
    return int64_to_int32(field_to_int64(col, name, truncate), name)


def from_structured(rec, truncate=False, rebase_t=False):
    """Structured x,y,t,p record array -> (n,4) int32.  ``truncate``: float fields are cut toward zero as
    the reference's astype does (else they must be integral); ``rebase_t``: t - t.min() in int64 first (what
    MixedDensityEventStack.stack does itself, mixed_density_event_stack.py:33), so absolute timestamps fit."""
    n = rec.shape[0]
    if rec.dtype == EVENT_DTYPE and rec.ndim == 1 and rec.flags.c_contiguous:
        # the adapters' own layout (four packed '<i4' fields, gen1_2yolo.py:567-571): a view, no per-field conversion
        # (rebase_t: nothing to do -- the timestamps ARE int32, and the one builder that asks for it, MixedDensityEventStack, only
        #  reads (t - t.min()) / (t.max() - t.min()), which its kernels form from 64-bit differences themselves; the strided min
        #  over the window was 9 us of a 100 us sample)
        return rec.view(np.int32).reshape(n, 4)
    ev = np.empty((n, 4), dtype=np.int32)
    for k, name in enumerate(("x", "y", "t", "p")):
        v = field_to_int64(rec[name], name, truncate)
        if name == "t" and rebase_t and n:
            v = v - v.min()
        ev[:, k] = int64_to_int32(v, name)
    return ev


# ----------------------------------------------------------------------------------------------
# Non-uniform streams (r04).  Real event windows are edge-clustered; the SURVEY 8(d) contract above (x, y ~ U) is the
# headline's workload, these two are what the sweep measures beside it (tools/bench_sweep.py, bench.py `sweep`).
# ----------------------------------------------------------------------------------------------
def _finish(x, y, t, n, rng, polarity, single_polarity=None):
    ev = np.empty((n, 4), dtype=np.int32)
    ev[:, 0], ev[:, 1] = x, y
    t = t.astype(np.int32)
    if n:
        t -= t[0]
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


def make_events_moving_circle(n, width, height, seed=0, span_us=50000, polarity="pm1", flow=(10.0, 0.0),
                              circle_radius=5.0, starting_point=(10.0, 10.0)):
    """The reference's own fake stream -- ev-licious ``generate_fake_events``
    (ev-licious/src/evlicious/io/utils/fake_events.py:5-29): every event lies on the rim of a circle that moves with
    the optical flow, ``x = u0 + t*vx + r cos(a)``, ``y = v0 + t*vy + r sin(a)``, ``t ~ sort(U[0,1))``, ``a ~ U[0, 2pi)``
    -- restated with a seeded generator and scaled from its 30x30 default frame to ``width x height`` (all lengths times
    ``min(width, height) / 30``), so the default flow (10, 0) sweeps a third of the frame and every event stays in frame
    (the reference masks out-of-frame events; none arise here for width >= height).  The rim's tangent rows collect ~4 %
    of the window each: a handful of 128-pixel units hold a thousand records while most of the frame is empty."""
    rng = np.random.default_rng(seed)
    s = min(width, height) / 30.0
    time = np.sort(rng.random(n))
    angle = rng.random(n) * 2.0 * np.pi
    u0, v0 = starting_point[0] * s, starting_point[1] * s
    x = (u0 + time * flow[0] * s + np.cos(angle) * circle_radius * s).astype(np.int64)
    y = (v0 + time * flow[1] * s + np.sin(angle) * circle_radius * s).astype(np.int64)
    np.clip(x, 0, width - 1, out=x)      # a caller's own flow / start may leave the frame: clamp instead of dropping,
    np.clip(y, 0, height - 1, out=y)     # so that the window keeps its n events
    return _finish(x, y, time * span_us, n, rng, polarity)


def make_events_edges(n, width, height, seed=0, span_us=50000, polarity="pm1", hot_fraction=0.8, hot_pixels=0.05,
                      n_edges=12, thickness=3):
    """Edge-cluster model: ``hot_fraction`` of the events fall (uniformly) on the pixels of ``n_edges`` straight edges of
    ``thickness`` pixels and random position / orientation that together cover ``hot_pixels`` of the frame, the rest is
    uniform background.  With the defaults 80 % of the events lie on 5 % of the pixels (2.6 events per edge pixel for a
    640x480 window of 50 000 events): near-horizontal edges give row chunks of several hundred records, near-vertical
    ones a few hot pixels in many chunks."""
    rng = np.random.default_rng(seed)
    target = int(hot_pixels * width * height)
    mask = np.zeros((height, width), dtype=bool)
    yy, xx = np.mgrid[0:height, 0:width]
    for k in range(n_edges):
        # edge k: the pixels within thickness/2 of a segment through a random point, random direction, of the length
        # that gives the edge its share of the target area
        cx, cy = rng.random() * width, rng.random() * height
        th = rng.random() * np.pi
        half = 0.5 * target / (n_edges * thickness)
        dx, dy = np.cos(th), np.sin(th)
        along = (xx - cx) * dx + (yy - cy) * dy
        across = -(xx - cx) * dy + (yy - cy) * dx
        mask |= (np.abs(along) <= half) & (np.abs(across) <= thickness / 2.0)
    hot = np.flatnonzero(mask.reshape(-1))
    n_hot = int(round(hot_fraction * n)) if hot.size else 0
    pix = np.concatenate([hot[rng.integers(0, max(hot.size, 1), size=n_hot)],
                          rng.integers(0, width * height, size=n - n_hot)])
    rng.shuffle(pix)
    t = np.sort(rng.integers(0, span_us, size=n))
    return _finish(pix % width, pix // width, t, n, rng, polarity)


GENERATORS = {"uniform": make_events, "circle": make_events_moving_circle, "edges": make_events_edges}
