"""Does a hot list that is too small get REPORTED?  The plan's total_events is shrunk after the workspace is allocated (hot_cap
is derived from it at launch time), clustered windows are built with a builder that defers hot units, and the status word is
read back: EVREP_ST_HOT_OVERFLOW must be set once units could not be queued."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from event_representation_study_amd import _lib, engine as eng
from event_representation_study_amd.synthetic import GENERATORS

H, W, N, B = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (240, 304, 50000, 32)))
wins = [GENERATORS["circle"](N, W, H, seed=900 + i) for i in range(B)]
eb = eng.EventBatch.from_numpy(wins, H, W)
for shrink in (None, 1):
    if shrink is not None:
        eb.plan.total_events = shrink
    eb.rebin()
    v = eb.voxel(5)
    torch.cuda.synchronize()
    hot = eb.workspace[eb.plan.off_scratch: eb.plan.off_scratch + 4 * 2048].view(torch.int32).cpu().numpy()
    st = eb.status()
    print("total_events", eb.plan.total_events, "status bits", sorted(set(int(s) for s in st)), "overflow windows",
          int(sum(1 for s in st if int(s) & _lib.ST_HOT_OVERFLOW)), "list counters after the call (must be 0):", int(np.abs(hot[::16]).sum()))
