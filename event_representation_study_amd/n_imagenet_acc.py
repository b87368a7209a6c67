"""Row F4: n_imagenet's per-polarity accumulators on the binned stream.

Host-side mirror of the `reshape_then_*` accumulator family of
n_imagenet/real_cnn_model/data/imagenet.py (same names, same `(event_tensor, augment=None, **kwargs)`
signature, same (C, H, W) float32 result, same empty-input behaviour):

    reshape_then_acc            :169-210   [pos count / max, pos latest time, neg count / max, neg latest time]
    reshape_then_acc_time       :213-247   [pos earliest, pos latest, neg earliest, neg latest]
    reshape_then_acc_count      :250-293   [pos count, pos latest, neg count, neg latest]
    reshape_then_acc_count_pol  :296-321   [pos count, neg count]
    reshape_then_acc_count_only :324-343   [count]
    reshape_then_acc_all        :346-394   [pos count, neg count, pos latest, neg latest, pos earliest, neg earliest]
    reshape_then_flat           :397-413   [any event]
    reshape_then_flat_pol       :416-438   [any pos event, any neg event]
    reshape_then_acc_exp        :441-472   exp(-(1 - latest)/0.3) per polarity, over the whole frame
    reshape_then_acc_time_pol   :475-510   [pos latest, neg latest]
    reshape_then_acc_intensity  :841-870   min-max normalised (pos count - neg count)

`event_tensor` is what parse_event hands over (:128-164): a float64 (N, 4) tensor of rows
[x, y, t_seconds, p], p in {-1, +1}, coordinates possibly fractional (they are truncated by `.long()`,
:187).  All of them are ONE HIP builder (evrep_polstats, csrc/evrep_builders.hip k_polstats) with a
per-channel (polarity class, statistic) descriptor; the two window-level normalisations (count / count.max(),
min-max of the intensity) are strided float32 tensor ops on the builder's output, as in the reference.

The three n_imagenet wrappers of the path's own builders that run in the reference -- reshape_then_optimized
(:1025-1039), reshape_then_event_stack (:1042-1060), reshape_then_tore (:1080-1107) -- are mirrored here too
(float64 event tensors with integral values, as load_event produces them without `reshape`).
reshape_then_time_surface (:1110-1137) raises IndexError in the reference itself (a float `p` field indexes
the surface memory, time_surface.py:67) and does so here; reshape_then_voxel_grid / _to_image go through tonic
(absent; see representations/tonic_compat.py).

reshape_then_acc_adj_sort (DiST, :873-999) = the same builder (per-polarity count, latest and earliest time)
followed by the reference's own image-space statements (count clipping, 5x5 pooling, temporal discount, dense
rank of the discounted timestamps) as torch ops on the GPU.

reshape_then_acc_sort (:513-838) is mirrored for strict=False (the "sorted timestamp image": latest time INDEX per
pixel, from the same builder with the dense time rank as the per-event value) and for strict=True (followed by the dense
rank of the per-pixel maxima; bit-exact on goldens that the generator checks not to depend on which of several tied events
torch_scatter's arg-max names); the denoise options raise the NameError they raise in the reference
(density_filter_event_image is never defined there).
The EST quantisation layer is in est.py.
The product path needs the HIP library and an MI355X; there is no CPU fallback.
"""
import numpy as np
import torch

from .engine import EventBatch

IMAGE_H = 224    # imagenet.py:17-18
IMAGE_W = 224
EXP_TAU = 0.3    # imagenet.py:20

ANY, POS, NEG = 0, 1, 2                                  # EVREP_PS_* of include/evrep.h
COUNT, TMAX, TMIN, FLAG, EXP, SIGNED = 0, 1, 2, 3, 4, 5

# name -> (polarity class per channel, statistic per channel)
SPECS = {
    "acc": ([POS, POS, NEG, NEG], [COUNT, TMAX, COUNT, TMAX]),
    "acc_time": ([POS, POS, NEG, NEG], [TMIN, TMAX, TMIN, TMAX]),
    "acc_count": ([POS, POS, NEG, NEG], [COUNT, TMAX, COUNT, TMAX]),
    "acc_count_pol": ([POS, NEG], [COUNT, COUNT]),
    "acc_count_only": ([ANY], [COUNT]),
    "acc_all": ([POS, NEG, POS, NEG, POS, NEG], [COUNT, COUNT, TMAX, TMAX, TMIN, TMIN]),
    "flat": ([ANY], [FLAG]),
    "flat_pol": ([POS, NEG], [FLAG, FLAG]),
    "acc_exp": ([POS, NEG], [EXP, EXP]),
    "acc_time_pol": ([POS, NEG], [TMAX, TMAX]),
    "acc_intensity": ([ANY], [SIGNED]),
}


def _as_f64(event_tensor):
    a = event_tensor.detach().cpu().numpy() if isinstance(event_tensor, torch.Tensor) else np.asarray(event_tensor)
    return np.asarray(a, dtype=np.float64).reshape(-1, 4)


def _window(ev, H, W):
    """(N,4) float64 -> int32 rows [x.long(), y.long(), 0, sign(p)] and the normalised float64 times
    (t - t[0]) / (t[-1] - t[0])  (imagenet.py:178-181,198-199)."""
    xi, yi = ev[:, 0].astype(np.int64), ev[:, 1].astype(np.int64)   # .long() truncates toward zero
    if len(ev) and (xi.min() < 0 or yi.min() < 0 or xi.max() >= W or yi.max() >= H):
        # the reference fails here as well (bincount grows past H*W and the reshape raises, :187-189)
        raise RuntimeError("event coordinates outside the %dx%d frame" % (W, H))
    rows = np.zeros((len(ev), 4), dtype=np.int32)
    rows[:, 0], rows[:, 1] = xi, yi
    rows[:, 3] = (ev[:, 3] > 0).astype(np.int32) - (ev[:, 3] < 0).astype(np.int32)
    with np.errstate(divide="ignore", invalid="ignore"):
        tn = (ev[:, 2] - ev[0, 2]) / (ev[-1, 2] - ev[0, 2]) if len(ev) else np.zeros(0)
    return rows, tn


def accumulate_batch(name, event_tensors, height=IMAGE_H, width=IMAGE_W, device="cuda:0"):
    """Batched form: a list of (N_b, 4) float64 event tensors -> (B, C, H, W) float32 CUDA tensor (a
    channel-first VIEW of the builder's (B, H, W, C) output, like the reference's permute(2, 0, 1))."""
    pol, stat = SPECS[name]
    wins = [_as_f64(e) for e in event_tensors]
    for w in wins:
        if len(w) == 0:
            raise IndexError("empty event tensor")   # event_tensor[0, 2], imagenet.py:178
    packed = [_window(w, height, width) for w in wins]
    batch = EventBatch.from_numpy([r for r, _ in packed], height, width, device=device)
    tnorm = torch.from_numpy(np.concatenate([t for _, t in packed]) if packed else np.zeros(0)).to(batch.device)
    out = batch.polstats(tnorm, pol, stat, tau=EXP_TAU)            # (B, H, W, C) float32
    if name == "acc":          # pos_count / pos_count.max().float()  (:190-191,196-197): float32 / float32
        for c in (0, 2):
            out[..., c] /= out[..., c].amax(dim=(1, 2), keepdim=True)
    elif name == "acc_intensity":   # (i - i.min()) / (i.max() - i.min())  (:867)
        lo = out.amin(dim=(1, 2, 3), keepdim=True)
        hi = out.amax(dim=(1, 2, 3), keepdim=True)
        out = (out - lo) / (hi - lo)
    return out.permute(0, 3, 1, 2)


def _single(name, event_tensor, augment, kwargs):
    if augment is not None:
        event_tensor = augment(event_tensor)
    H = kwargs.get("height", IMAGE_H)
    W = kwargs.get("width", IMAGE_W)
    ev = _as_f64(event_tensor)
    if len(ev) == 0:
        if name in ("acc_count", "acc_time_pol"):   # ten synthetic events at the origin (:258-261,483-486)
            ev = np.zeros((10, 4))
            ev[:, 2] = (np.arange(10, dtype=np.float32) / np.float32(10.0)).astype(np.float64)
            ev[:, 3] = 1
        elif name == "acc_all":                     # :353-354 (IMAGE_H x IMAGE_W whatever height/width say)
            return torch.zeros([6, IMAGE_H, IMAGE_W])
    res = accumulate_batch(name, [ev], H, W, device=kwargs.get("device", "cuda:0"))[0]
    return res if kwargs.get("keep_on_device", False) else res.cpu()


def reshape_then_acc(event_tensor, augment=None, **kwargs):
    return _single("acc", event_tensor, augment, kwargs)


def reshape_then_acc_time(event_tensor, augment=None, **kwargs):
    return _single("acc_time", event_tensor, augment, kwargs)


def reshape_then_acc_count(event_tensor, augment=None, **kwargs):
    return _single("acc_count", event_tensor, augment, kwargs)


def reshape_then_acc_count_pol(event_tensor, augment=None, **kwargs):
    return _single("acc_count_pol", event_tensor, augment, kwargs)


def reshape_then_acc_count_only(event_tensor, augment=None, **kwargs):
    return _single("acc_count_only", event_tensor, augment, kwargs)


def reshape_then_acc_all(event_tensor, augment=None, **kwargs):
    return _single("acc_all", event_tensor, augment, kwargs)


def reshape_then_flat(event_tensor, augment=None, **kwargs):
    return _single("flat", event_tensor, augment, kwargs)


def reshape_then_flat_pol(event_tensor, augment=None, **kwargs):
    return _single("flat_pol", event_tensor, augment, kwargs)


def reshape_then_acc_exp(event_tensor, augment=None, **kwargs):
    return _single("acc_exp", event_tensor, augment, kwargs)


def reshape_then_acc_time_pol(event_tensor, augment=None, **kwargs):
    return _single("acc_time_pol", event_tensor, augment, kwargs)


def reshape_then_acc_intensity(event_tensor, augment=None, **kwargs):
    return _single("acc_intensity", event_tensor, augment, kwargs)


CLIP_COUNT_RATE = 0.99   # imagenet.py:23
DISC_ALPHA = 3.0         # imagenet.py:24


def reshape_then_acc_adj_sort(event_tensor, augment=None, **kwargs):
    """DiST (imagenet.py:873-999): discounted, sorted timestamp image, (2, H, W) float32."""
    F = torch.nn.functional
    if augment is not None:
        event_tensor = augment(event_tensor)
    H = kwargs.get("height", IMAGE_H)
    W = kwargs.get("width", IMAGE_W)
    # [pos count, pos latest, pos earliest, neg count, neg latest, neg earliest] from the HIP builder
    pol, stat = [POS, POS, POS, NEG, NEG, NEG], [COUNT, TMAX, TMIN, COUNT, TMAX, TMIN]
    ev = _as_f64(event_tensor)
    if len(ev) == 0:
        raise IndexError("empty event tensor")
    rows, tn = _window(ev, H, W)
    batch = EventBatch.from_numpy([rows], H, W, device=kwargs.get("device", "cuda:0"))
    prim = batch.polstats(torch.from_numpy(tn).to(batch.device), pol, stat)[0]          # (H, W, 6)
    halves = []
    for k in (0, 3):
        count, out, min_out = prim[..., k].clone(), prim[..., k + 1].clone(), prim[..., k + 2].clone()
        # clip count (:893-901)
        unique_count = torch.unique(count, return_counts=True)[1]
        sum_subset = torch.cumsum(unique_count, dim=0)
        th_clip = sum_subset[sum_subset < H * W * CLIP_COUNT_RATE].shape[0]
        count[count > th_clip] = th_clip
        min_out[count == 0] = 1.0                                                        # (:924-925)
        patch = 5
        neighbor = patch ** 2 * F.avg_pool2d(count.unsqueeze(0), patch, stride=1, padding=patch // 2)
        disc = (F.max_pool2d(out.unsqueeze(0), patch, stride=1, padding=patch // 2)
                + F.max_pool2d(-min_out.unsqueeze(0), patch, stride=1, padding=patch // 2)) / neighbor
        out[count > 0] = out[count > 0] - DISC_ALPHA * disc.squeeze()[count > 0]         # (:957-961)
        out[out < 0] = 0
        out[neighbor.squeeze() == 1.0] = 0
        flat = out.reshape(H * W)
        val, idx = torch.sort(flat)                                                      # (:973-990)
        unq, cnt = torch.unique_consecutive(val, return_counts=True)
        srt = torch.zeros_like(flat)
        srt[idx] = torch.repeat_interleave(torch.arange(unq.shape[0], device=flat.device), cnt).float() / unq.shape[0]
        halves.append(srt.reshape(H, W))
    res = torch.stack(halves, dim=2).permute(2, 0, 1).float()
    return res if kwargs.get("keep_on_device", False) else res.cpu()


TIME_SCALE = 1000000     # imagenet.py:21


def reshape_then_acc_sort(event_tensor, augment=None, **kwargs):
    """Sorted timestamp image (imagenet.py:513-838): per pixel (and polarity) the LATEST time index (strict=False) or
    the dense rank of it among the pixels' latest indices (strict=True), optionally with the binary event image and
    quantised copies; (C, H, W) float32."""
    if augment is not None:
        event_tensor = augment(event_tensor)
    global_time, neglect, use_image, strict = (kwargs[k] for k in ("global_time", "neglect_polarity", "use_image", "strict"))
    H = kwargs.get("height", IMAGE_H)
    W = kwargs.get("width", IMAGE_W)
    ev_t = event_tensor if isinstance(event_tensor, torch.Tensor) else torch.as_tensor(np.asarray(event_tensor))
    time_idx = (ev_t[:, 2] * TIME_SCALE).long()
    if global_time:   # dense rank of the microsecond timestamps (:521-525)
        mem, cnt = torch.unique_consecutive(time_idx, return_counts=True)
        time_idx = torch.repeat_interleave(torch.arange(mem.shape[0]), cnt)
    ev_t[:, 2] = time_idx                      # the reference overwrites the caller's time column as well (:525,539)
    if use_image and kwargs["denoise_image"]:
        raise NameError("name 'density_filter_event_image' is not defined")   # what the reference raises (:556,679)
    if kwargs["denoise_sort"]:
        raise NameError("name 'density_filter_event_image' is not defined")   # (:609,777)
    ev = _as_f64(ev_t)
    if len(ev) == 0:
        raise RuntimeError("max(): Expected reduction dim to be specified for input.numel() == 0")
    rows, _ = _window(ev, H, W)
    batch = EventBatch.from_numpy([rows], H, W, device=kwargs.get("device", "cuda:0"))
    tval = torch.from_numpy(np.ascontiguousarray(ev[:, 2])).to(batch.device)       # float64 time index per event
    classes = [ANY] if neglect else [POS, NEG]
    pol = [c for c in classes for _ in (0, 1)]
    stat = [FLAG, TMAX] * len(classes)
    prim = batch.polstats(tval, pol, stat)[0]                                      # (H, W, 2 * len(classes))
    q = kwargs["quantize_sort"]
    chans = []
    for k in range(len(classes)):
        image, srt = prim[..., 2 * k], prim[..., 2 * k + 1]
        if strict:
            # :563-590,685-748: per pixel the event of the latest time index (scatter_max's arg; which of several
            # events sharing pixel AND index it names does not matter, they carry the same value and the stream is
            # time-ordered), then the DENSE RANK of those per-pixel maxima (+1), min-max normalised in float32
            hot = image > 0
            if not bool(hot.any()):
                # a polarity without events is replaced by ONE event at pixel (0, 0), t = 0 (:647-652): its rank
                # image is all zero (max == min -> fill_(0)), its event image has that one pixel set
                image = torch.zeros_like(image)
                image[0, 0] = 1.0
                srt = torch.zeros_like(srt)
            else:
                vals = srt[hot]
                uniq, inv = torch.unique(vals, sorted=True, return_inverse=True)
                fs = inv.to(torch.float32) + 1
                if int(uniq.numel()) > 1:
                    fs = (fs - fs.min()) / (fs.max() - fs.min())
                else:
                    fs = torch.zeros_like(fs)
                srt = torch.zeros_like(srt)
                srt[hot] = fs
        elif not bool((srt > 0.0).any()):      # hot_event_sort.max() on an empty selection (:592-594,744-746)
            raise RuntimeError("max(): Expected reduction dim to be specified for input.numel() == 0")
        if q is not None:
            if type(q) == int:
                srt = torch.round(srt * q) / q
            elif type(q) == list:
                srt = torch.stack([torch.round(srt * qs) / qs for qs in q], dim=2)
        parts = ([image.unsqueeze(-1)] if use_image else []) + [srt if srt.dim() == 3 else srt.unsqueeze(-1)]
        chans.append(torch.cat(parts, dim=2))
    res = torch.cat(chans, dim=2).permute(2, 0, 1).float()
    return res if kwargs.get("keep_on_device", False) else res.cpu()


# ---------------------------------------------------------------------------------------------
# n_imagenet's wrappers of the path's own builders (imagenet.py:1002-1137)
# ---------------------------------------------------------------------------------------------
def fix_events_training(events):
    """(N, 4) float array -> structured array with '<f8' fields x, y, t, p (imagenet.py:1002-1006)."""
    ev = np.ascontiguousarray(np.asarray(events, dtype=np.float64).reshape(-1, 4))
    return ev.view([("x", "<f8"), ("y", "<f8"), ("t", "<f8"), ("p", "<f8")]).reshape(-1)


def _prep(event_tensor, augment, kwargs):
    if augment is not None:
        event_tensor = augment(event_tensor)
    return fix_events_training(_as_f64(event_tensor)), kwargs.get("height", IMAGE_H), kwargs.get("width", IMAGE_W)


def reshape_then_optimized(event_tensor, augment=None, **kwargs):
    from .representations.optimized_representation import get_optimized_representation
    data, H, W = _prep(event_tensor, augment, kwargs)
    rep = get_optimized_representation(data, data.shape[0], H, W)
    return torch.tensor(rep.transpose(2, 0, 1)).float()


def reshape_then_event_stack(event_tensor, augment=None, **kwargs):
    from .representations.event_stack import EventStack
    data, H, W = _prep(event_tensor, augment, kwargs)
    data["p"] = (data["p"] + 1) // 2
    transformation = EventStack(12, data.shape[0], H, W)
    post = transformation.post_stack(transformation.pre_stack(data, data[-1]["t"]))
    return torch.tensor(post.transpose(3, 0, 1, 2)[..., 0]).float()


def reshape_then_tore(event_tensor, augment=None, **kwargs):
    from .representations.tore import events2ToreFeature
    data, H, W = _prep(event_tensor, augment, kwargs)
    x, y, ts, pol = data["x"], data["y"], data["t"], data["p"]
    x = x - min(x) + 1          # the reference's shift to 1-based coordinates (:1095-1096)
    y = y - min(y) + 1
    rep = events2ToreFeature(x, y, ts, pol, ts[-1], 6, (H, W))
    return torch.tensor(rep.transpose(2, 0, 1)).float()


def reshape_then_time_surface(event_tensor, augment=None, **kwargs):
    # the reference stores int8 polarities back into a '<f8' field and then indexes memory[p, y, x] with it
    raise IndexError("only integers, slices (`:`), ellipsis (`...`), numpy.newaxis (`None`) and integer or boolean "
                     "arrays are valid indices")
