import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from event_representation_study_amd import engine as eng
from event_representation_study_amd.synthetic import GENERATORS
oracle.build()
W, H, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (304, 240, 50000)))
dist = sys.argv[4] if len(sys.argv) > 4 else "circle"
ev = GENERATORS[dist](N, W, H, seed=100)
eb = eng.EventBatch.from_numpy([ev], H, W)
print("pass", eb.plan.reserved)
nch = (W + 127) // 128
key = ev[:, 1] * nch + ev[:, 0] // 128
cnt = np.bincount(key, minlength=H * nch).reshape(H, nch)
def report(name, got, ref):
    bad = np.argwhere(~((got == ref) | ((got != got) & (ref != ref))))
    print(name, "mismatches", len(bad))
    if len(bad):
        rows = {}
        for y, x, c in bad[:200000]:
            rows.setdefault((y, x // 128), []).append((x, c))
        for (y, ck), lst in list(rows.items())[:12]:
            xs = sorted(set(x for x, c in lst))
            print("  unit row %d chunk %d nrec %d: %d bad px, x range %d..%d, first %r got %r ref %r" % (
                y, ck, cnt[y, ck], len(xs), xs[0], xs[-1], lst[0], got[y, lst[0][0], lst[0][1]], ref[y, lst[0][0], lst[0][1]]))
report("ergo12", eb.optimized()[0].cpu().numpy(), oracle.ergo12(ev, H, W))
report("event_stack", eb.event_stack()[0].cpu().numpy(), oracle.event_stack(ev, H, W))
report("voxel", eb.voxel(5)[0].cpu().numpy(), oracle.voxel(ev, H, W, 5))
