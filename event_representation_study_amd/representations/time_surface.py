"""ToTimesurface -- mirrors representations/time_surface.py:7-74 of the reference."""
from dataclasses import dataclass
from typing import Tuple, Union

import numpy as np

from ._common import events_from_fields, raise_for_status
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
        if len(indices) == 0:
            return np.zeros((0, self.sensor_size[2], H, W))
        ev = events_from_fields(events["x"], events["y"], events["t"], events["p"])
        batch = EventBatch.from_numpy(ev, H, W)
        raise_for_status(batch, what="ToTimesurface")
        out = np.empty((len(indices), 2, H, W), dtype=np.float64)
        for s0 in range(0, len(indices), 8):                       # 8 slices per launch
            chunk = indices[s0:s0 + 8]
            rep = batch.time_surface(slices=len(chunk), tau=float(self.tau), premap=False, indices=chunk)
            rep = rep[0].cpu().numpy().reshape(H, W, len(chunk), 2)  # channel c = 2*s + p
            out[s0:s0 + len(chunk)] = rep.transpose(2, 3, 0, 1)
            if s0 + 8 < len(indices) and not _chain_alive(chunk, ev.shape[0]):
                out[s0 + 8:] = 0.0
                break
        return out


def _chain_alive(chunk, n):
    return all(b > a for a, b in zip(chunk, chunk[1:])) and chunk[-1] < n
