"""Stand-ins for the two ``tonic.transforms`` classes the reference dispatcher instantiates
(gen1_transforms.py:21-25, :44-49).  tonic itself is an un-vendored, unpinned third-party
dependency of the reference (absent here), so these follow tonic's published algorithms and their
parity is UNPINNED (SURVEY.md section 8(c), rows A2' / A8').  ``str(cls)`` contains the substrings
the dispatcher greps for ("ToVoxelGrid", "ToImage")."""
import numpy as np

from ._common import finish, sample_batch


class ToVoxelGrid:
    """tonic.transforms.ToVoxelGrid(sensor_size=(W,H,2), n_time_bins): (T, 1, H, W) float64 grid with
    bilinear interpolation in time, polarity 0 counted as -1."""

    def __init__(self, sensor_size, n_time_bins):
        self.sensor_size = sensor_size
        self.n_time_bins = int(n_time_bins)

    def build_hwt(self, events, scale=1.0, device_out=False):
        """(H, W, T) float64 on the host, `scale` folded into the kernel (what the dispatcher wants)."""
        W, H = int(self.sensor_size[0]), int(self.sensor_size[1])
        if self.n_time_bins > 16:
            raise NotImplementedError("ToVoxelGrid: n_time_bins > 16")
        sb = sample_batch(events, H, W, device_out=device_out)
        return finish(sb, sb.voxel(bins=self.n_time_bins, mode=1, scale=float(scale)), what="ToVoxelGrid")

    def __call__(self, events):
        # tonic's layout (T, 1, H, W), as a strided view of the builder's (H, W, T) result
        return np.moveaxis(self.build_hwt(events), -1, 0)[:, np.newaxis, :, :]


class ToImage:
    """tonic.transforms.ToImage(sensor_size=(W,H,2)): one (2, H, W) int16 frame of per-polarity event
    counts (polarity values 0 and 1 index the two channels)."""

    def __init__(self, sensor_size):
        self.sensor_size = sensor_size

    def _build(self, events, device_out):
        W, H = int(self.sensor_size[0]), int(self.sensor_size[1])
        sb = sample_batch(events, H, W, device_out=device_out)
        # counts of p == 0 ("count_neg" falls back to p == 0 when no -1 is present) and of p == 1
        import torch
        rep = sb.mdes([0, 0], ["count_neg", "count_pos"], ["sum", "sum"], dtype=torch.float32)
        frames = torch.empty((1, 2, H, W), dtype=torch.int16, device=rep.device)
        frames[0].copy_(rep[0].permute(2, 0, 1))                                # one kernel: transpose + cast (counts are exact in float32)
        return finish(sb, frames, what="ToImage")

    def __call__(self, events):
        return self._build(events, False)

    def build_cuda(self, events):
        """The (2, H, W) int16 frame as a CUDA tensor (get_item_transform_cuda)."""
        return self._build(events, True)
