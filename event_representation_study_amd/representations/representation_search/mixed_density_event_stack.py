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
        if self.stacking_type != "SBN":
            raise NotImplementedError("only the 'SBN' stacking the reference selects is implemented")
        windows, funcs, aggs = self.indexes_functions_aggregations
        sb = sample_batch(event_sequence, self.height, self.width, truncate=True, rebase_t=True)   # astype + t - t.min(), :26-33
        from ... import _lib
        def window(v):      # anything that cannot index the 7-window list fails the channel (-> zeros)
            ok = isinstance(v, (int, np.integer)) and not isinstance(v, bool) and -7 <= int(v) <= 6
            return (int(v) % 7) if ok else None
        w = [window(v) for v in list(windows)[: self.stack_size]]
        f = [v if v in _lib.FUNCS else None for v in list(funcs)[: self.stack_size]]
        a = [v if v in _lib.AGGS else None for v in list(aggs)[: self.stack_size]]
        return finish(sb, sb.mdes(w, f, a, scale=1.0), allow_oob=True, what="MixedDensityEventStack")
