"""Store pacing of the builders (NOTES.md 3.2, Store pacing): the same launch into several ~1 GB output tensors alive at once, for a list of
hold values (10 ns ticks; 0 = unpaced).  Which of the allocations are "slow" placements shows in the first column.

    python tools/experiments/pacing.py [builder] [nbuf] [hold ...]
    builder: ergo64 (default) | ergo32 | ts64 | voxel | ergo64_1mpx | ts64_1mpx
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from event_representation_study_amd._lib import check
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import make_events

which = sys.argv[1] if len(sys.argv) > 1 else "ergo64"
nbuf = int(sys.argv[2]) if len(sys.argv) > 2 else 6
holds = [int(v) for v in sys.argv[3:]] or [0, 500, 550, 600, 650, 700, 750]
if which.endswith("_1mpx"):
    H, W, N, B = 720, 1280, 200000, 8
else:
    H, W, N, B = 480, 640, 50000, 32
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
eb.bin()


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


kind = which.split("_")[0]
C, dt = {"ergo64": (12, torch.float64), "ergo32": (12, torch.float32), "ts64": (12, torch.float64),
         "voxel": (5, torch.float64)}[kind]
outs = [torch.empty((B, H, W, C), dtype=dt, device="cuda:0") for _ in range(nbuf)]


def launch(o):
    if kind == "ergo64":
        eb.optimized(out=o)
    elif kind == "ergo32":
        eb.optimized(dtype=torch.float32, out=o)
    elif kind == "ts64":
        eb.time_surface(out=o)
    else:
        eb.voxel(out=o)


print("# %s  %dx%d  %d windows x %d events; columns = hold in 10 ns ticks; us per launch" % (which, W, H, B, N))
print("%-14s" % "buffer" + "".join(" %7d" % h for h in holds))
for o in outs:
    row = []
    for h in holds:
        check(eb.lib.evrep_plan_set_pacing(ctypes.byref(eb.plan), h), "evrep_plan_set_pacing")
        row.append(t(lambda: launch(o)))
    print("%-14x" % o.data_ptr() + "".join(" %7.1f" % v for v in row), flush=True)
