"""events2ToreFeature -- mirrors representations/tore.py:6-83 of the reference."""
import numpy as np

from ._common import events_from_fields, raise_for_status
from ..engine import EventBatch


def events2ToreFeature(x, y, ts, pol, sampleTimes, k, frameSize):
    """TORE volume for one sample time.  x, y are 1-based (the reference indexes ``[i - 1, j - 1]``);
    returns (frameSize[0], frameSize[1], 2k) float32: the k most recent log-intervals per polarity."""
    Hf, Wf = int(frameSize[0]), int(frameSize[1])
    x = np.asarray(x)
    y = np.asarray(y)
    if len(x) and (x.min() < 1 or y.min() < 1):
        raise NotImplementedError("events2ToreFeature: x, y below 1 (numpy negative-index wrap) are not supported")
    ev = events_from_fields(x.astype(np.int64) - 1, y.astype(np.int64) - 1, ts, pol)
    if ev.shape[0] == 0:
        out = np.zeros((Hf, Wf, 2 * k), dtype=np.float32)
        out[...] = np.float32(np.log(np.float32(500e6) + 1) - np.log(151))
        return out
    batch = EventBatch.from_numpy(ev, Hf, Wf)
    raise_for_status(batch, what="events2ToreFeature")
    rep = batch.tore(k=int(k), frame_mode=2, scale=1.0, sample_times=[int(sampleTimes)])
    return rep[0].cpu().numpy()
