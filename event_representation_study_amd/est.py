"""Row F4 (second half): the EST quantisation layer of ev-YOLOv6, forward only.

Mirror of ``QuantizationLayer.forward`` (ev-YOLOv6/yolov6/models/learned_repr.py:143-179): events
``(N, 5)`` rows ``[x, y, t, p, b]`` -> per batch item, polarity and time bin i the image
``sum_events t_n * f(t_n - i/(C-1))`` with ``t_n = t / t.max()`` (:159-160) and ``f`` the layer's value MLP
(1 -> 100 -> 100 -> 1, LeakyReLU(0.1), :9-43), channels ``[p*C + i]`` (:175-176), letterboxed to
``image_size`` with bilinear interpolation and the constant 114 (:93-136).

The MLP is a scalar function of a scalar with piecewise-linear activations, hence EXACTLY piecewise linear:
``PiecewiseLinearKernel`` derives its breakpoints in closed form from the weights (the kinks of the first
layer, then the zero crossings of every second-layer pre-activation inside each first-layer piece) and the
HIP builder (k_est) evaluates ``f`` by a bucketed breakpoint search + one fused multiply-add in float64 --
~10^4 multiply-adds per (event, bin) become a table lookup, and the layer is a per-pixel segmented
reduction on the binned stream like every other builder.  Forward only: training the layer needs autograd
and stays with the reference.

Differences, all on the safe side: pixel/voxel indices are exact integers (the reference forms them in
float32, :163, which loses bits beyond 2**24 voxels, i.e. for more than ~19 batch items of 6x240x304); p must
be in {0, 1} (a p of -1 makes the reference clamp negative indices onto voxel 0, :170); the caller's events
tensor is not modified (the reference normalises ``t`` in place through a view, :156-160).
"""
import numpy as np
import torch

from .engine import EventBatch

NEG_SLOPE = 0.1


def mlp_weights(value_layer):
    """(w1, b1, W2, b2, w3, b3) float64 arrays of a ValueLayer-like module (``.mlp`` = three nn.Linear) or of a
    state dict with keys mlp.{0,1,2}.{weight,bias}."""
    sd = value_layer if isinstance(value_layer, dict) else value_layer.state_dict()
    g = lambda k: np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], dtype=np.float64)  # noqa: E731
    w1, b1 = g("mlp.0.weight").reshape(-1), g("mlp.0.bias").reshape(-1)
    W2, b2 = g("mlp.1.weight"), g("mlp.1.bias").reshape(-1)
    w3, b3 = g("mlp.2.weight").reshape(-1), float(g("mlp.2.bias").reshape(-1)[0])
    if W2.shape != (len(b2), len(w1)) or len(w3) != len(b2):
        raise ValueError("expected a 1 -> n -> m -> 1 MLP")
    return w1, b1, W2, b2, w3, b3


def _leaky(z, slope=NEG_SLOPE):
    return np.where(z > 0, z, slope * z)


class PiecewiseLinearKernel:
    """f(u) = w3 . leaky(W2 leaky(w1 u + b1) + b2) + b3 on [lo, hi] as sorted breakpoints + per-piece (a, c)."""

    def __init__(self, weights, lo=-1.0, hi=1.0, nbucket=4096, slope=NEG_SLOPE):
        w1, b1, W2, b2, w3, b3 = weights
        self.weights, self.lo, self.hi, self.slope = weights, float(lo), float(hi), float(slope)
        with np.errstate(divide="ignore", invalid="ignore"):
            k1 = -b1 / w1                                     # kinks of the first layer
        k1 = k1[np.isfinite(k1) & (k1 > lo) & (k1 < hi)]
        edges = np.concatenate([[lo], np.sort(k1), [hi]])
        kinks = [k1]
        for a, b in zip(edges[:-1], edges[1:]):                # inside (a, b) the first layer is linear in u
            if not b > a:
                continue
            mid = 0.5 * (a + b)
            alpha = np.where(w1 * mid + b1 > 0, 1.0, slope)    # which side of its kink each neuron is on
            A = W2 @ (alpha * w1)                              # z2_k(u) = A_k u + B_k on this piece
            Bc = W2 @ (alpha * b1) + b2
            with np.errstate(divide="ignore", invalid="ignore"):
                z = -Bc / A
            kinks.append(z[np.isfinite(z) & (z > a) & (z < b)])
        bp = np.unique(np.concatenate(kinks))
        self.edges = np.concatenate([[lo], bp, [hi]])          # piece k = [edges[k], edges[k+1])
        mids = 0.5 * (self.edges[:-1] + self.edges[1:])
        # slope and intercept of every piece from the activation pattern at its midpoint (exact: no kink inside)
        a1 = np.where(np.outer(mids, w1) + b1 > 0, 1.0, slope)             # (pieces, n1)
        A2 = (a1 * w1) @ W2.T                                              # d z2 / du
        B2 = (a1 * b1) @ W2.T + b2
        a2 = np.where(A2 * mids[:, None] + B2 > 0, 1.0, slope)
        self.a = (a2 * A2) @ w3
        self.c = (a2 * B2) @ w3 + b3
        self.nbucket = int(nbucket)
        left = lo + (hi - lo) * np.arange(self.nbucket) / self.nbucket
        self.bucket = (np.searchsorted(self.edges[1:], left, side="right")).astype(np.int32)  # first piece whose end > left edge
        self.bucket = np.minimum(self.bucket, len(self.a) - 1).astype(np.int32)
        self._dev = {}

    def __len__(self):
        return len(self.a)

    def __call__(self, u):
        """float64 evaluation through the table (host; what the HIP builder computes)."""
        u = np.asarray(u, dtype=np.float64)
        k = np.clip(np.searchsorted(self.edges[1:-1], u, side="right"), 0, len(self.a) - 1)
        return self.a[k] * u + self.c[k]

    def mlp(self, u):
        """The MLP itself in float64 (ground truth for the table)."""
        w1, b1, W2, b2, w3, b3 = self.weights
        u = np.asarray(u, dtype=np.float64).reshape(-1)
        h1 = _leaky(np.outer(u, w1) + b1, self.slope)
        h2 = _leaky(h1 @ W2.T + b2, self.slope)
        return h2 @ w3 + b3

    def device_table(self, device):
        key = str(device)
        if key not in self._dev:
            seg = np.stack([self.edges[1:], self.a, self.c], axis=1)       # {u_next, a, c}
            self._dev[key] = (torch.from_numpy(np.ascontiguousarray(seg)).to(device),
                              torch.from_numpy(self.bucket).to(device))
        return self._dev[key]


def letterbox_image_batch(image_batch, size, color=114):
    """learned_repr.py:93-136: bilinear resize with unchanged aspect ratio, centred on a `color` canvas."""
    bsz, c, orig_h, orig_w = image_batch.shape
    scale = min(size / orig_w, size / orig_h)
    new_w, new_h = int(orig_w * scale), int(orig_h * scale)
    resized = torch.nn.functional.interpolate(image_batch, size=(new_h, new_w), mode="bilinear", align_corners=False)
    canvas = torch.full((bsz, c, size, size), fill_value=color, dtype=image_batch.dtype, device=image_batch.device)
    top, left = (size - new_h) // 2, (size - new_w) // 2
    canvas[:, :, top:top + new_h, left:left + new_w] = resized
    return canvas


class QuantizationLayer:
    """Forward-only mirror of learned_repr.QuantizationLayer(dim=(C, H, W), image_size)."""

    def __init__(self, dim, value_layer, image_size=640, device="cuda:0"):
        self.dim = tuple(int(v) for v in dim)
        if not 2 <= self.dim[0] <= 8:
            raise ValueError("2 <= C <= 8 bins (2C channels <= 16)")
        self.image_size = image_size
        self.device = torch.device(device)
        self.kernel = value_layer if isinstance(value_layer, PiecewiseLinearKernel) else \
            PiecewiseLinearKernel(mlp_weights(value_layer))

    def voxel(self, events):
        """(N, 5) [x, y, t, p, b] -> (B, 2C, H, W) float32 on the device, before the letterbox (:143-176)."""
        C, H, W = self.dim
        ev = events.detach().to("cpu", torch.float32) if isinstance(events, torch.Tensor) else \
            torch.as_tensor(np.asarray(events, dtype=np.float32))
        if ev.dim() != 2 or ev.shape[1] != 5 or ev.shape[0] == 0:
            raise ValueError("events must be a non-empty (N, 5) tensor of [x, y, t, p, b] rows")
        b = ev[:, 4].to(torch.int64)
        nb = int(1 + ev[-1, 4].item())                                   # B = 1 + events[-1, -1]  (:145)
        if bool((b[1:] < b[:-1]).any()):
            raise NotImplementedError("events must be grouped by batch index (ascending), as the collate function delivers them")
        p = ev[:, 3]
        if bool(((p != 0) & (p != 1)).any()):
            raise ValueError("p must be in {0, 1} (learned_repr.py:163 indexes the polarity half with it)")
        counts = torch.bincount(b, minlength=nb)[:nb]
        offs = np.zeros(nb + 1, dtype=np.int64)
        np.cumsum(counts.numpy(), out=offs[1:])
        tn = ev[:, 2].clone()
        for bi in range(nb):                                             # t[b == bi] /= t[b == bi].max()  (:159-160)
            s, e = int(offs[bi]), int(offs[bi + 1])
            if e > s:
                tn[s:e] /= tn[s:e].max()
        rows = np.zeros((ev.shape[0], 4), dtype=np.int32)
        rows[:, 0] = ev[:, 0].numpy().astype(np.int64)                   # idx.long() truncates  (:170)
        rows[:, 1] = ev[:, 1].numpy().astype(np.int64)
        rows[:, 3] = p.numpy().astype(np.int32)
        if rows[:, 0].min() < 0 or rows[:, 1].min() < 0 or rows[:, 0].max() >= W or rows[:, 1].max() >= H:
            raise IndexError("event coordinates outside the %dx%d frame" % (W, H))
        batch = EventBatch(torch.from_numpy(rows).to(self.device), torch.from_numpy(offs), H, W)
        seg, bucket = self.kernel.device_table(self.device)
        out = batch.est_voxel(tn.contiguous().to(self.device), C, seg, bucket, self.kernel.lo, self.kernel.hi)
        return out.permute(0, 3, 1, 2)                                   # (B, 2C, H, W): [p*C + i]  (:175-176)

    def forward(self, events):
        vox = self.voxel(events)
        if self.image_size is None:
            return vox.contiguous()
        return letterbox_image_batch(vox.contiguous(), self.image_size).to(dtype=torch.float32)

    __call__ = forward
