"""EventStack -- mirrors representations/event_stack.py:5-131 of the reference.

``pre_stack`` launches the HIP path (binning + k_event_stack); ``post_stack`` hands back the dense
stack in the reference's layout.  Only time-sorted input with ``last_timestamp >= t[-1]`` (what both
reference callers pass: gen1_transforms.py:37-39, n_imagenet imagenet.py:1053-1057) is supported;
a non-empty "future" half raises NotImplementedError.
"""
import numpy as np

from ._common import events_from_fields, raise_for_status
from ..engine import EventBatch


class EventStack(object):
    NO_VALUE = 0.0
    STACK_LIST = ["stacked_polarity", "index"]

    def __init__(self, stack_size, num_of_event, height, width):
        self.stack_size = stack_size
        self.num_of_event = num_of_event
        self.height = height
        self.width = width

    def pre_stack(self, event_sequence, last_timestamp):
        x, y = event_sequence["x"], event_sequence["y"]
        p, t = event_sequence["p"], event_sequence["t"]
        assert len(x) == len(y) == len(p) == len(t)
        t64 = np.asarray(t).astype(np.int64)
        if np.any(t64 > last_timestamp):
            raise NotImplementedError("EventStack: events after last_timestamp (the 'future' half) are not supported")
        ev = events_from_fields(x, y, t64, np.asarray(p).astype(np.int8))   # p is {0,1} here; the kernel forms 2p-1
        batch = EventBatch.from_numpy(ev, self.height, self.width)
        raise_for_status(batch, what="EventStack")
        dense = batch.event_stack(self.stack_size, premap=False, scale=1.0)  # (1, H, W, S) float32
        return [{"dense": dense}]

    def post_stack(self, pre_stacked_event):
        dense = pre_stacked_event[0]["dense"][0].cpu().numpy()               # (H, W, S)
        return dense[:, :, np.newaxis, :]                                    # (H, W, 1, S), event_stack.py:61-63
