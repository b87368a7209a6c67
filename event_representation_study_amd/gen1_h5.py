"""The event side of the reference's Gen1 container (ev-YOLOv6/yolov6/data/gen1_2yolo.py:72-82,150-198): one HDF5 group per
recording, ``<name>/events/{x, y, t, p, height, width}`` and ``<name>/bbox/{t_unique, event_idx, ...}``; sample ``idx`` is the
window of the ``num_events`` events in front of the idx-th labelled timestamp of the recordings taken in name order
(``convert_idx_to_rel_idx``, ``_load_bbox``: event_idx; ``_load_events``: [max(0, event_idx - num_events), event_idx), t rebased
to the window's first event).  Read with h5lite -- chunk-wise, Blosc included -- so a window costs its own chunks, not the
recording.  ``windows(indices)`` hands (n, 4) int32 arrays to ``EventBatch.from_numpy`` / the precompute pipeline."""
import numpy as np

from .synthetic import int64_to_int32

from . import h5lite


class Gen1H5Events:
    def __init__(self, path, num_events=50000):
        self.h5 = h5lite.File(str(path))
        self.num_events = int(num_events)
        self._file_names = sorted(self.h5.keys())                                              # :77
        self._num_unique_bboxes = [len(self.h5["%s/bbox/t_unique" % f]) for f in self._file_names]   # :78-80
        first = self._file_names[0]
        self.height = int(self.h5["%s/events/height" % first][()])                             # :82-83
        self.width = int(self.h5["%s/events/width" % first][()])
        self._event_idx = {}

    def __len__(self):
        return int(sum(self._num_unique_bboxes))

    def locate(self, idx):
        """(index inside its recording, recording name): convert_idx_to_rel_idx, :158-166"""
        if not 0 <= idx < len(self):
            raise IndexError(idx)
        counter = 0
        while idx >= self._num_unique_bboxes[counter]:
            idx -= self._num_unique_bboxes[counter]
            counter += 1
        return idx, self._file_names[counter]

    def window(self, idx):
        """(n, 4) int32 rows [x, y, t - t[0], p] of sample idx (_load_bbox's event_idx, _load_events)."""
        rel, name = self.locate(idx)
        if name not in self._event_idx:
            self._event_idx[name] = np.asarray(self.h5["%s/bbox/event_idx" % name][:]).astype(np.int64)
        idx1 = int(self._event_idx[name][rel])
        idx0 = max(0, idx1 - self.num_events)
        ev = self.h5["%s/events" % name]
        x, y, t, p = (np.asarray(ev[k][idx0:idx1]) for k in ("x", "y", "t", "p"))
        if idx1 - idx0 <= 0:
            raise IndexError("sample %d: no events before its label (the reference fails on xyt[0, -1], gen1_2yolo.py:196)" % idx)
        out = np.empty((idx1 - idx0, 4), dtype=np.int32)
        # range-checked narrowing: a corrupt container (or a window of more than 2^31 us) fails loudly instead of wrapping
        out[:, 0], out[:, 1], out[:, 3] = (int64_to_int32(np.asarray(v).astype(np.int64), n) for v, n in ((x, "x"), (y, "y"), (p, "p")))
        out[:, 2] = int64_to_int32(t.astype(np.int64) - int(t[0]), "t")                         # xyt[:, -1] -= xyt[0, -1], :196
        return out

    def windows(self, indices):
        return [self.window(int(i)) for i in indices]
