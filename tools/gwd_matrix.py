#!/usr/bin/env python3
"""BASELINE.json configs[3]: the GWD score C_p of every named representation over a set of windows
(the reference's `gen1_compute.py` loop), windows dealt to the ranks, scalars assembled with one
all_gather.  Runs on 1 GPU as is; under `python -m torch.distributed.run --nproc-per-node N` on N.

Synthetic Gen1-shaped windows (304x240, 50 000 events): n ~ 12.5k events and m <= 14.4k representation
points per quadrant solve, 3 solves per (representation, window).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=4)
    ap.add_argument("--events", type=int, default=50000)
    ap.add_argument("--host-harness", action="store_true", help="r02's path (host quadrant bookkeeping, one solve per call)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl")

    from event_representation_study_amd import gwd_pipeline as gp
    from event_representation_study_amd.engine import EventBatch
    from event_representation_study_amd.synthetic import make_events

    H, W, S = 240, 304, 240
    wins = [make_events(args.events, W, H, seed=3000 + i) for i in range(args.windows)]
    dev = torch.device("cuda", torch.cuda.current_device())

    reps = {  # the six representations of gen1_compute.py:117-124, built as gen1_transforms drives them (x255), one launch
              # per representation over this rank's windows
        "VoxelGrid": lambda b: b.voxel(12, mode=1, scale=255.0),
        "MixedDensityEventStack": lambda b: b.optimized(scale=255.0),
        "EventStack": lambda b: b.event_stack(12, premap=True, scale=255.0),
        "TimeSurface": lambda b: b.time_surface(6, 50000.0, premap=True, scale=255.0),
        "2DHistogram": lambda b: b.mdes([0, 0], ["count_neg", "count_pos"], ["sum", "sum"], scale=255.0),
        "TORE": lambda b: b.tore(6, frame_mode=0, scale=255.0),
    }
    if args.host_harness:   # r02's path: per (representation, window) a host harness and three single solves
        def eb(ev):
            return EventBatch.from_numpy(ev, H, W, device=dev)
        one = {k: (lambda ev, f=f: (lambda r: r[0] if not isinstance(r, list) else r[0])(f(eb(ev))).to(torch.float64)) for k, f in reps.items()}
    gp.measure_cp_device(wins, reps, H, W, S)   # warm, same shapes: first launches, scratch allocations (hipMalloc), tap tables
    out = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.host_harness:
        for name, build in one.items():
            out[name] = gp.measure_cp(wins, build, H, W, S)[0]
    else:
        out = {k: v[0] for k, v in gp.measure_cp_device(wins, reps, H, W, S).items()}
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"C_p": out, "windows": args.windows, "events_per_window": args.events, "n_gpus": world,
                          "solves": 3 * args.windows * len(reps), "wall_s": el,
                          "harness": "host" if args.host_harness else "device"}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
