"""MixedDensityEventStack -- mirrors representation_search/mixed_density_event_stack.py:8-151."""
import numpy as np

from .._common import finish, sample_batch


class MixedDensityEventStack(object):
    def __init__(self, stack_size, num_of_events, height, width, indexes_functions_aggregations, stacking_type):
        self.stack_size = stack_size
        self.num_of_events = num_of_events
        self.height = height
        self.width = width
        self.indexes_functions_aggregations = indexes_functions_aggregations
        self.stacking_type = stacking_type

    def stack(self, event_sequence):
        """(H, W, stack_size) float64.  A channel whose triple is unusable (``None`` window, unknown
        function, out-of-frame events in its window) is all zeros, like the reference's try/except."""
        # "SBN": 7 windows cut by event count (:58-74); "SBT": 8 windows cut by normalised time (:76-107).  Any other string
        # leaves the reference with the one full window, so every index but 0 / -1 fails its channel.
        nwin = {"SBN": 7, "SBT": 8}.get(self.stacking_type, 1)
        windows, funcs, aggs = self.indexes_functions_aggregations
        sb = sample_batch(event_sequence, self.height, self.width, truncate=True, rebase_t=True)   # astype + t - t.min(), :26-33
        from ... import _lib
        def window(v):      # anything that cannot index the window list fails the channel (-> zeros)
            ok = isinstance(v, (int, np.integer)) and not isinstance(v, bool) and -nwin <= int(v) <= nwin - 1
            return (int(v) % nwin) if ok else None
        w = [window(v) for v in list(windows)[: self.stack_size]]
        f = [v if v in _lib.FUNCS else None for v in list(funcs)[: self.stack_size]]
        a = [v if v in _lib.AGGS else None for v in list(aggs)[: self.stack_size]]
        return finish(sb, sb.mdes(w, f, a, scale=1.0, stacking="SBT" if self.stacking_type == "SBT" else "SBN"), allow_oob=True,
                      what="MixedDensityEventStack", allow_unsorted=self.stacking_type != "SBT")   # "SBT" cuts by time: rank ranges need order
