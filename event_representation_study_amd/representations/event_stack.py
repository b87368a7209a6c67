"""EventStack -- mirrors representations/event_stack.py:5-131 of the reference.

``pre_stack`` splits the events at ``last_timestamp`` exactly as the reference does (event_stack.py:21-41):
the "past" half (``t <= last_timestamp``) is stacked as it is, the "future" half (``t > last_timestamp``) is
reversed and its polarity negated.  Each half is one window of a two-window device batch (binning pass +
k_event_stack); ``post_stack`` hands the dense levels back in the reference's layout, the future half with its
level axis reversed (:64-65).  Stacking only depends on the order of the events, never on their timestamps
(``t_s`` is computed and dropped by the reference, :69-78), so the device batch carries no time at all.
"""
import numpy as np

from ._common import raise_for_status
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
        # the reference's own casts (event_stack.py:16-19): float fields are truncated, p wraps as int8 does
        x = np.asarray(event_sequence["x"]).astype(np.int32)
        y = np.asarray(event_sequence["y"]).astype(np.int32)
        p = 2 * np.asarray(event_sequence["p"]).astype(np.int8) - 1
        t = np.asarray(event_sequence["t"]).astype(np.int64)
        assert len(x) == len(y) == len(p) == len(t)

        past = t <= last_timestamp
        future = t > last_timestamp
        halves = [(x[past], y[past], p[past])]
        if np.sum(future) != 0:
            halves.append((x[future][::-1], y[future][::-1], p[future][::-1] * -1))
        wins = []
        for hx, hy, hp in halves:
            ev = np.zeros((len(hx), 4), dtype=np.int32)      # t column stays 0: order is all that matters
            ev[:, 0], ev[:, 1], ev[:, 3] = hx, hy, hp
            wins.append(ev)
        batch = EventBatch.from_numpy(wins, self.height, self.width)
        raise_for_status(batch, what="EventStack", any_window_oob=True)   # an empty past half: p_t.min() raises (:24)
        dense = batch.event_stack(self.stack_size, premap=2, scale=1.0)   # (1 or 2, H, W, S) float32
        return [{"dense": dense[i]} for i in range(len(wins))]

    def post_stack(self, pre_stacked_event):
        levels = [h["dense"].cpu().numpy() for h in pre_stacked_event]       # each (H, W, S)
        if len(levels) == 2:
            levels[1] = levels[1][:, :, ::-1]                                # event_stack.py:64-65
        return np.stack(levels, axis=2)                                      # (H, W, 1 or 2, S), :66-68
