"""``get_item_transform`` -- drop-in for representations/gen1_transforms.py:12-89 of the reference.

Same contract: the representation is chosen by substring tests on ``representation_name`` (the
``str()`` of whatever the caller keeps in its name->class table), hyper-parameters are fixed at the
dispatch site (12 bins / 12 stack levels / k = 6 / tau = 50000 / 6 slices), three branches rewrite
the caller's ``["p"]`` field in place, and the result is scaled by 255.  Here every in-repo branch is
one fused launch sequence on the GPU (binning pass + builder, the x255 folded into the kernel).
"""
import numpy as np

from ._common import UnsortedWindow, finish, sample_batch

STACK_LEVELS = 12      # gen1_transforms.py:35
VOXEL_BINS = 12        # :22
TORE_K = 6             # :52
TS_TAU = 50000         # :76
TS_SLICES = 6          # :81
SCALE = 255


def _third_party(events, transform, height, width, **kw):
    # tonic-style transforms are classes taking sensor_size=(W, H, 2); ours live in tonic_compat
    return transform((width, height, 2), **kw)(events)


def _voxel_grid(events, transform, height, width, num_events, device_out=False):
    made = transform((width, height, 2), n_time_bins=VOXEL_BINS)
    if hasattr(made, "build_hwt"):       # our ToVoxelGrid: (H, W, T) straight from the builder, x255 in the kernel
        return made.build_hwt(events, scale=SCALE, device_out=device_out)
    if device_out:
        raise NotImplementedError("device_out needs this package's ToVoxelGrid (a third-party transform returns host arrays)")
    grid = made(events)                                                             # (T, 1, H, W)
    return np.moveaxis(grid[:, 0], 0, -1) * SCALE                                   # (H, W, T)


def _optimized(events, transform, height, width, num_events, device_out=False):
    sb = sample_batch(events, height, width, truncate=True, rebase_t=True, device_out=device_out)   # MDES.stack's own casts (:26-33)
    return finish(sb, sb.optimized(scale=float(SCALE)), allow_oob=True, what="MixedDensityEventStack", allow_unsorted=True)


def _event_stack(events, transform, height, width, num_events, device_out=False):
    # EventStack.pre_stack splits at last_timestamp = t[-1] (:36-38, event_stack.py:23,27) and the dispatcher keeps the past
    # half (:40); the stack itself only looks at the ORDER of the events (put is last-write-wins in array order,
    # event_stack.py:125).  On ascending timestamps the past half is the whole window; otherwise it is the events with
    # t <= t[-1], in array order (r04: unsorted windows are built in array order, as the reference does)
    t = np.asarray(events["t"])
    sb = sample_batch(events, height, width, device_out=device_out)    # the whole window is on its way while the host looks for a future half
    past = None
    if len(t) and (t.dtype.kind == "f" or t.max() > t[-1]):      # (integer timestamps: one reduction says whether a future half exists at all)
        past = t.astype(np.int64) <= t[-1] if t.dtype.kind == "f" else t <= t[-1]   # pre_stack compares t.astype(int64) (event_stack.py:36-38)
    if past is not None and not past.all():
        sb = sample_batch(events[past], height, width, device_out=device_out)   # (unsorted timestamps only: staged again without it)
    dev = sb.event_stack(STACK_LEVELS, premap=True, scale=float(SCALE))
    events["p"] = (events["p"] + 1) // 2                      # side effect the reference has (:34) -- while the GPU works (the events were staged above)
    return finish(sb, dev, what="EventStack", allow_unsorted=True)


def _histogram(events, transform, height, width, num_events, device_out=False):
    frames = transform((width, height, 2))
    events["p"] = (events["p"] + 1) // 2                      # (:46)
    if device_out:
        if not hasattr(frames, "build_cuda"):
            raise NotImplementedError("device_out needs this package's ToImage (a third-party transform returns host arrays)")
        return frames.build_cuda(events).permute(1, 2, 0) * SCALE
    img = np.moveaxis(frames(events), 0, -1)
    img *= SCALE
    return img


def _tore(events, transform, height, width, num_events, device_out=False):
    # (timestamps that are not ascending: array order, the reference's np.partition on its k-vector -- tore.py:22-25, k_tore)
    sb = sample_batch(events, height, width, device_out=device_out)
    # bounding-box frame, origin-shifted, sample time t[-1] (:61-66): frame_mode 0; the box travels with the result
    return finish(sb, sb.tore_full(k=TORE_K, frame_mode=0, scale=float(SCALE)), what="TORE", tore_k=TORE_K, allow_unsorted=True)


def _time_surface(events, transform, height, width, num_events, device_out=False):
    sb = sample_batch(events, height, width, device_out=device_out)
    t = np.asarray(events["t"])       # (read before the side effect below; the events themselves are on their way already)
    # the six cuts searchsorted(t_norm, 1..6) are taken on the device from the same float64 formula
    dev = sb.time_surface(TS_SLICES, float(TS_TAU), premap=1, scale=float(SCALE))
    events["p"] = ((events["p"] + 1) / 2).astype(np.int8)     # (:70-72) -- while the GPU works
    try:
        return finish(sb, dev, what="ToTimesurface")
    except UnsortedWindow:
        # timestamps not ascending (r04): the reference's scan runs in array order whatever they are (time_surface.py:66-74),
        # and so do the kernels; the cuts are what the DISPATCHER's own numpy call yields on such an array (:79-81) -- taken
        # here, handed over as explicit indices -- and the exponentials are not factorised (premap bit 1)
        t_norm = (t - t[0]) / (t[-1] - t[0]) * TS_SLICES
        idx = np.searchsorted(t_norm, np.arange(TS_SLICES) + 1)
        sb.batch._binned = True       # the same events, already binned
        return finish(sb, sb.time_surface(TS_SLICES, float(TS_TAU), premap=3, scale=float(SCALE), indices=idx),
                      what="ToTimesurface", allow_unsorted=True)


# order matters: "MixedDensityEventStack" contains "EventStack" (:27 is tested before :33)
_BRANCHES = (
    ("ToVoxelGrid", False, _voxel_grid),
    ("MixedDensityEventStack", False, _optimized),
    ("EventStack", False, _event_stack),
    ("ToImage", False, _histogram),
    ("TORE", True, _tore),                # matched case-insensitively: the repr of events2ToreFeature
    ("ToTimesurface", False, _time_surface),
)


def get_item_transform(reshaped_return_data, representation_name, transform, height, width, num_events,
                       time_window):
    for needle, fold_case, build in _BRANCHES:
        if needle in (representation_name.upper() if fold_case else representation_name):
            return build(reshaped_return_data, transform, height, width, num_events)
    raise UnboundLocalError("local variable 'rep' referenced before assignment")   # what the reference does


def get_item_transform_cuda(reshaped_return_data, representation_name, transform, height, width, num_events,
                            time_window=None):
    """``get_item_transform`` with the result left on the GPU: the same dispatch, side effects, x255 and exceptions, but
    a fresh ``(H, W, C)`` CUDA tensor instead of a numpy array -- no 29.5 MB float64 read-back (0.53 of a sample's
    0.63 ms).  For callers that move the representation to the device anyway, as the reference's trainer does right after
    the dataset (``Trainer.prepro_data``, ev-YOLOv6/yolov6/core/engine.py:629-635: ``.to(device).float() / 255``)."""
    for needle, fold_case, build in _BRANCHES:
        if needle in (representation_name.upper() if fold_case else representation_name):
            return build(reshaped_return_data, transform, height, width, num_events, device_out=True)
    raise UnboundLocalError("local variable 'rep' referenced before assignment")
