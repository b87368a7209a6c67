#!/usr/bin/env python3
"""Soak: every builder (and the binning pass under it) run ITERS times on the same resident batches must give
bit-identical tensors every time (no race between LDS phases, no order dependence in the partition).

    python tools/soak_determinism.py [iters]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_representation_study_amd.engine import EventBatch, gwd_padded_l1  # noqa: E402
from event_representation_study_amd.synthetic import make_events  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    cfgs = [(32, 50000, 480, 640), (8, 500000, 480, 640), (4, 200000, 720, 1280), (16, 3000, 48, 64)]
    bad = 0
    for B, N, H, W in cfgs:
        wins = [make_events(N, W, H, seed=100 * B + i) for i in range(B)]
        if B == 16:                         # hot pixel + one-row windows in the small configuration
            wins[0][:, 0], wins[0][:, 1] = 5, 7
            wins[1][:, 1] = 3
        eb = EventBatch.from_numpy(wins, H, W)
        tn = torch.rand(eb.total, dtype=torch.float64, device=eb.device)
        builders = {
            "optimized": lambda: eb.optimized(), "optimized_f32": lambda: eb.optimized(dtype=torch.float32),
            "event_stack": lambda: eb.event_stack(), "time_surface": lambda: eb.time_surface(),
            "tore": lambda: eb.tore(6, frame_mode=2), "voxel": lambda: eb.voxel(5),
            "mdes_rt": lambda: eb.mdes([0, 4, 2, 6, 1], ["timestamp", "count", "polarity", "timestamp_pos", "count_neg"],
                                       ["variance", "sum", "mean", "max", "mean"]),
            "polstats": lambda: eb.polstats(tn, [1, 2, 1, 2, 0], [0, 0, 1, 2, 4]),
        }
        ref = {}
        for it in range(iters):
            eb.rebin()
            for name, fn in builders.items():
                out = fn()
                key = out.view(torch.uint8).reshape(-1)
                if name not in ref:
                    ref[name] = key.clone()
                elif not torch.equal(key, ref[name]):
                    bad += 1
                    print("MISMATCH", (B, N, H, W), name, "iteration", it, int((key != ref[name]).sum()), "bytes differ")
        print("config", (B, N, H, W), "ok" if not bad else "FAILED", iters, "iterations x", len(builders), "builders")
    rng = np.random.default_rng(1)
    Xs = torch.from_numpy(rng.random((12500, 4))).cuda()
    Xt = torch.from_numpy(rng.random((14400, 14)) * 255).cuda()
    c0 = gwd_padded_l1(Xs, Xt).clone()
    for it in range(iters):
        if not torch.equal(gwd_padded_l1(Xs, Xt), c0):
            bad += 1
            print("MISMATCH gwd iteration", it)
    print("gwd", "ok" if not bad else "FAILED")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
