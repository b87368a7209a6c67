"""Convenience aliases named in BASELINE.json's north_star.  The reference has no
``Representation.construct`` and no ``gwd`` symbol; these are thin
names over the engine for callers who want tensors back without touching numpy."""
import numpy as np
import torch

from .engine import EventBatch, gwd_padded_l1
from .representations.representation_search.compute_otmi import otmi


class Representation:
    """Base: ``construct(events, H, W) -> torch.Tensor`` of shape (H, W, C) on the current GPU.
    ``events`` is an (n, 4) int array/tensor of [x, y, t, p] rows (time-sorted)."""

    name = "Representation"

    def _build(self, batch):
        raise NotImplementedError

    def construct(self, events, H, W):
        return self.construct_batch([events], H, W)[0]

    def construct_batch(self, windows, H, W, device="cuda:0"):
        arrs = [w.cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w) for w in windows]
        out = self._build(EventBatch.from_numpy(arrs, H, W, device=device))
        return out if isinstance(out, list) else out

    def __repr__(self):
        return "<%s>" % self.name


class OptimizedRepresentation(Representation):
    name = "MixedDensityEventStack/ERGO-12"

    def _build(self, batch):
        return batch.optimized()


class MixedDensityEventStack(Representation):
    name = "MixedDensityEventStack"

    def __init__(self, windows, functions, aggregations):
        self.triples = (list(windows), list(functions), list(aggregations))

    def _build(self, batch):
        return batch.mdes(*self.triples)


class EventStack(Representation):
    name = "EventStack"

    def __init__(self, stack_size=12):
        self.stack_size = stack_size

    def _build(self, batch):
        return batch.event_stack(self.stack_size, premap=True)


class TimeSurface(Representation):
    name = "ToTimesurface"

    def __init__(self, slices=6, tau=50000.0):
        self.slices, self.tau = slices, tau

    def _build(self, batch):
        return batch.time_surface(self.slices, self.tau, premap=True)


class ToRE(Representation):
    name = "TORE"

    def __init__(self, k=6, full_frame=True):
        self.k, self.full_frame = k, full_frame

    def _build(self, batch):
        return batch.tore(self.k, frame_mode=2 if self.full_frame else 0)


class VoxelGrid(Representation):
    name = "ToVoxelGrid"

    def __init__(self, bins=5):
        self.bins = bins

    def _build(self, batch):
        return batch.voxel(self.bins, mode=0)


def gwd(events, rep, height, width, rep_size):
    """The reference's GWD between a raw event window and one representation (compute_otmi.otmi)."""
    if isinstance(rep, torch.Tensor):
        rep = rep.detach().cpu().numpy()
    return float(otmi(events, rep, height, width, rep_size))


def gwd_point_clouds(Xs, Xt, h=0.7):
    return float(gwd_padded_l1(Xs, Xt, h).item())
