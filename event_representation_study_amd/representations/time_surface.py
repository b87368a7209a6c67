"""ToTimesurface -- mirrors representations/time_surface.py:7-74 of the reference."""
from dataclasses import dataclass
from typing import Tuple, Union

import numpy as np
import torch

from ._common import events_from_fields, raise_for_status
from .. import _lib
from ..engine import EventBatch


@dataclass(frozen=True)
class ToTimesurface:
    """Global exponential time surfaces sampled at the event indices ``indices``.

    sensor_size: (W, H, 2).  ``surface_dimensions`` and ``decay`` are accepted and ignored, exactly as
    the reference ignores them (time_surface.py:25-49)."""

    sensor_size: Tuple[int, int, int]
    surface_dimensions: Union[None, Tuple[int, int]] = None
    tau: float = 5e3
    decay: str = "lin"

    def __call__(self, events, indices):
        W, H = int(self.sensor_size[0]), int(self.sensor_size[1])
        indices = [int(i) for i in np.asarray(indices).reshape(-1)]
        out = np.zeros((len(indices), self.sensor_size[2], H, W))
        if len(indices) == 0:
            return out
        # timestamps of any dtype (time_surface.py:66-74 only stores and subtracts them): integral ones ride in the int32 event
        # rows, others as a float64 array of their own (r04) -- the rows' t column then only carries the order
        t = np.asarray(events["t"])
        tf = None
        if t.dtype.kind == "f" and len(t) and not np.all(t == np.rint(t)):
            if not np.all(np.isfinite(t)):
                raise OverflowError("non-finite timestamps")
            tf = np.ascontiguousarray(t, dtype=np.float64)
            asc = bool(np.all(tf[1:] >= tf[:-1]))
            t = np.arange(len(tf)) if asc else -np.arange(len(tf))     # order marker: the status word reports "not ascending"
        ev = events_from_fields(events["x"], events["y"], t, events["p"])
        batch = EventBatch.from_numpy(ev, H, W)
        tf_dev = None if tf is None else torch.from_numpy(tf).to(batch.device)
        # the scan of time_surface.py:66-74 runs in array order: timestamps that are not ascending only forbid the factorised
        # exponentials (premap bit 1)
        unsorted = bool(raise_for_status(batch, what="ToTimesurface", allow_unsorted=True) & _lib.ST_UNSORTED)
        # the reference's scan tests `index == indices[pos]` once per event: only a strictly increasing
        # prefix of in-range indices is ever reached, every later surface stays all-zero
        live, prev = 0, -1
        for i in indices:
            if not (prev < i < ev.shape[0]):
                break
            live, prev = live + 1, i
        for s0 in range(0, live, 8):                                  # up to 8 surfaces per launch
            chunk = indices[s0:min(s0 + 8, live)]
            rep = batch.time_surface(slices=len(chunk), tau=float(self.tau), premap=2 if unsorted else 0, indices=chunk, times_f64=tf_dev)
            batch.check_built("ToTimesurface")                        # the hot-list flag is raised BY the builder
            rep = rep[0].cpu().numpy().reshape(H, W, len(chunk), 2)    # channel c = 2*s + p
            out[s0:s0 + len(chunk)] = rep.transpose(2, 3, 0, 1)
        return out
