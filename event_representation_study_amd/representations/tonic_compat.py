"""Stand-ins for the two ``tonic.transforms`` classes the reference dispatcher instantiates
(gen1_transforms.py:21-25, :44-49).  tonic itself is an un-vendored, unpinned third-party
dependency of the reference (absent here), so these follow tonic's published algorithms and their
parity is UNPINNED (SURVEY.md section 8(c), rows A2' / A8').  ``str(cls)`` contains the substrings
the dispatcher greps for ("ToVoxelGrid", "ToImage")."""
import numpy as np

from ._common import raise_for_status, single_batch


class ToVoxelGrid:
    """tonic.transforms.ToVoxelGrid(sensor_size=(W,H,2), n_time_bins): (T, 1, H, W) float64 grid with
    bilinear interpolation in time, polarity 0 counted as -1."""

    def __init__(self, sensor_size, n_time_bins):
        self.sensor_size = sensor_size
        self.n_time_bins = int(n_time_bins)

    def __call__(self, events):
        W, H = int(self.sensor_size[0]), int(self.sensor_size[1])
        batch = single_batch(events, H, W)
        raise_for_status(batch, what="ToVoxelGrid")
        T = self.n_time_bins
        parts = [batch.voxel(bins=T, mode=1)] if T <= 16 else None
        if parts is None:
            raise NotImplementedError("ToVoxelGrid: n_time_bins > 16")
        rep = parts[0][0].cpu().numpy()                      # (H, W, T)
        return np.ascontiguousarray(rep.transpose(2, 0, 1))[:, np.newaxis, :, :]


class ToImage:
    """tonic.transforms.ToImage(sensor_size=(W,H,2)): one (2, H, W) int16 frame of per-polarity event
    counts (polarity values 0 and 1 index the two channels)."""

    def __init__(self, sensor_size):
        self.sensor_size = sensor_size

    def __call__(self, events):
        W, H = int(self.sensor_size[0]), int(self.sensor_size[1])
        batch = single_batch(events, H, W)
        raise_for_status(batch, what="ToImage")
        # counts of p == 0 ("count_neg" falls back to p == 0 when no -1 is present) and of p == 1
        rep = batch.mdes([0, 0], ["count_neg", "count_pos"], ["sum", "sum"])[0].cpu().numpy()
        return np.ascontiguousarray(rep.transpose(2, 0, 1)).astype(np.int16)
