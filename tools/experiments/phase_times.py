"""Where a builder wave's time goes (experiment build: -DEVREP_TIMING, EVREP_LIB_PATH=<that .so>).  Phase marks of k_mdes, mean
us since the wave started: 0 records fetched and grouped (unit_front), 1 window statistics merged, 2 segment heads listed +
digest, 3 reduced and published (pace entry), 4 pace left, 5 stores issued, 6 first loads issued, 7 sparse: reduced / dense: first part about to be reduced.

    hipcc ... -DEVREP_TIMING -o /tmp/libevrep_timing.so ...; EVREP_LIB_PATH=/tmp/libevrep_timing.so python tools/experiments/phase_times.py [hold ...]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from event_representation_study_amd._lib import check
from event_representation_study_amd.engine import EventBatch
from event_representation_study_amd.synthetic import GENERATORS

H, W, N, B = 480, 640, 50000, 32
if os.environ.get("SHAPE"):   # SHAPE=W,H,N,B: another workload (dense windows: 640,480,500000,8; marks 3/4 = first part tile)
    W, H, N, B = (int(v) for v in os.environ["SHAPE"].split(","))
DIST = os.environ.get("DIST", "uniform")   # uniform | circle | edges (synthetic.GENERATORS)
eb = EventBatch.from_numpy([GENERATORS[DIST](N, W, H, seed=i) for i in range(B)], H, W)
eb.bin()
DT = torch.float32 if os.environ.get("DTYPE") == "f32" else torch.float64   # DTYPE=f32: the float32 ERGO-12 instance
BUILDER = os.environ.get("BUILDER", "optimized")                              # BUILDER=voxel: k_voxel's marks (r06)
CH = 5 if BUILDER == "voxel" else 12


def build(o):
    if BUILDER == "voxel":
        eb.voxel(5, out=o)
    else:
        eb.optimized(out=o, dtype=DT)


outs = [torch.empty((B, H, W, CH), dtype=DT, device="cuda:0") for _ in range(int(os.environ.get("NBUF", "4")))]
holds = [int(v) for v in sys.argv[1:]] or [0, 600, 670]
nunit = B * H * ((W + 127) // 128)
assert nunit * 64 <= (eb.total + 1) * 8, "the idle half of the record stream is too small for the marks"
idle = eb.plan.off_sorted1 + (((eb.total * 8 + 255) // 256) * 256 if eb.plan.reserved == 2 else 0)   # see bin_view (EVREP_TIMING)
dbg = eb.workspace[idle: idle + nunit * 64].view(torch.int64).view(nunit, 8)
for o in outs:
    for h in holds:
        check(eb.lib.evrep_plan_set_pacing(ctypes.byref(eb.plan), h), "pacing")
        for _ in range(3):
            build(o)
        torch.cuda.synchronize()
        dbg.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            build(o)
        b.record()
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().astype(float) / 100.0
        import numpy as np
        ph = ["%d: %.2f/%.2f" % (i, d[:, i].mean(), np.percentile(d[:, i], 95)) for i in range(8)]
        life = d[:, 5][d[:, 5] > 0]      # mark 5 = the wave's last store is issued: its lifetime (hot items: of the item's wave)
        print("%x %s hold %4d  %.1f us/launch  wave lifetime mean %.2f p95 %.2f p99 %.2f max %.2f us  phases mean/p95 (us) %s"
              % (o.data_ptr(), DIST, h, a.elapsed_time(b) * 100, life.mean(), np.percentile(life, 95), np.percentile(life, 99), life.max(),
                 "  ".join(ph)), flush=True)
        if os.environ.get("TOP"):   # the longest-lived waves, mark by mark (hot items: 6 loads issued, 2 count sweep done,
            for i in np.argsort(-d[:, 5])[:int(os.environ["TOP"])]:   # 3 part in the slot, 0 front end done, 1 statistics, 4 staged, 5 stored)
                print("   unit %6d  " % i + "  ".join("%d: %.1f" % (k, d[i, k]) for k in (6, 2, 3, 0, 1, 4, 5)), flush=True)
