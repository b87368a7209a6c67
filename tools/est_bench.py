#!/usr/bin/env python3
"""EST quantisation layer forward (row F4) at the reference's shape: dim = (6, 240, 304) (yolo.py:57-59),
32 batch items x 50 000 events.  Build time of k_est (binning excluded and included), events/s.

    python tools/est_bench.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_representation_study_amd.engine import EventBatch  # noqa: E402
from event_representation_study_amd.est import PiecewiseLinearKernel  # noqa: E402
from event_representation_study_amd.synthetic import make_events  # noqa: E402


def trained_trilinear(C, steps=1000):
    """A 1 -> 100 -> 100 -> 1 LeakyReLU(0.1) MLP fitted to the trilinear kernel max(0, 1 - |u| (C - 1)) the way the
    reference initialises its value layer (learned_repr.py:45-70: Adam, lr 1e-2, 2000 uniform samples per step)."""
    torch.manual_seed(1)
    lin = [torch.nn.Linear(1, 100), torch.nn.Linear(100, 100), torch.nn.Linear(100, 1)]
    params = [p for l in lin for p in l.parameters()]
    opt = torch.optim.Adam(params, lr=1e-2)
    act = torch.nn.LeakyReLU(0.1)
    for _ in range(steps):
        u = torch.empty(2000, 1).uniform_(-1, 1)
        gt = torch.clamp(1 - u.abs() * (C - 1), min=0)
        pred = lin[2](act(lin[1](act(lin[0](u)))))
        loss = ((pred - gt) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    g = lambda t: t.detach().numpy().astype(np.float64)  # noqa: E731
    return (g(lin[0].weight).reshape(-1), g(lin[0].bias), g(lin[1].weight), g(lin[1].bias), g(lin[2].weight).reshape(-1),
            float(lin[2].bias.item()))


def main():
    C, H, W, B, N = 6, 240, 304, 32, 50000
    rng = np.random.default_rng(5)
    # a random 1 -> 100 -> 100 -> 1 MLP (PyTorch's default Linear init ranges): far more kinks than a trained kernel
    w1, b1 = rng.uniform(-1, 1, 100), rng.uniform(-1, 1, 100)
    W2, b2 = rng.uniform(-0.1, 0.1, (100, 100)), rng.uniform(-0.1, 0.1, 100)
    w3, b3 = rng.uniform(-0.1, 0.1, 100), 0.01
    kernels = {"random 1-100-100-1 MLP (default Linear init ranges)": PiecewiseLinearKernel((w1, b1, W2, b2, w3, b3)),
               "MLP trained to the trilinear kernel (1000 Adam steps, what ValueLayer.init_kernel does)":
                   PiecewiseLinearKernel(trained_trilinear(C))}
    wins = [make_events(N, W, H, seed=i, polarity="01") for i in range(B)]
    eb = EventBatch.from_numpy(wins, H, W)
    tn = torch.cat([torch.from_numpy((w[:, 2] / max(1, w[:, 2].max())).astype(np.float32)) for w in wins]).cuda()
    out = torch.empty((B, H, W, 2 * C), dtype=torch.float32, device=eb.device)

    def timed(fn, k=30):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / k
    t_bin = timed(lambda: eb.rebin())
    macs = B * N * C * (100 + 100 * 100 + 100)
    for name, kern in kernels.items():
        seg, bucket = kern.device_table(eb.device)
        t_est = timed(lambda: eb.est_voxel(tn, C, seg, bucket, kern.lo, kern.hi, out=out))
        print(json.dumps({"kernel": name, "dim": [C, H, W], "batch": B, "events_per_item": N, "pieces": len(kern),
                          "bin_ms": round(t_bin, 4), "est_ms": round(t_est, 4),
                          "events_per_s": round(B * N / ((t_bin + t_est) * 1e-3)),
                          "mlp_macs_replaced": macs,
                          "equivalent_TFLOPs": round(2 * macs / (t_est * 1e-3) / 1e12, 1)}))

if __name__ == "__main__":
    main()
