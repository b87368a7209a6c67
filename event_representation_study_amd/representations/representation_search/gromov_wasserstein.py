"""compute_repr -- mirrors representation_search/gromov_wasserstein.py:72-82 (the in-repo voxel grid)."""
import numpy as np

from .._common import raise_for_status
from ...engine import EventBatch


def compute_repr_from_events(events, width, height, bins=5):
    """Voxel grid of an (n, 4) int [x, y, t, p] window with the caller-side normalisation
    t = (t - t[0]) / (t[-1] - t[0]) of gromov_wasserstein.py:96 -> (H, W, bins) float64."""
    batch = EventBatch.from_numpy(np.asarray(events), height, width)
    raise_for_status(batch, what="compute_repr")
    return batch.voxel(bins=bins, mode=0)[0].cpu().numpy()
