"""The GWD legs under rocprofv3: one batched call of 144 solves (bench.py's leg) and 36 single solves (r02's path)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from event_representation_study_amd.engine import gwd_padded_l1, gwd_padded_l1_batch

rng = np.random.default_rng(77)
n, m, P = 12500, 14400, int(sys.argv[1]) if len(sys.argv) > 1 else 144
dev = torch.device("cuda:0")
Xs = torch.from_numpy(rng.random((n, 4))).to(dev)
Xt = torch.from_numpy(rng.random((m, 14)) * np.array([255.0] * 12 + [1.0, 1.0])).to(dev)
nn = torch.full((P,), n, dtype=torch.int64, device=dev)
mm = torch.full((P,), m, dtype=torch.int64, device=dev)
zero = torch.zeros(P, dtype=torch.int64, device=dev)
out = torch.zeros(P, dtype=torch.float64, device=dev)
for _ in range(2):
    gwd_padded_l1_batch(Xs, nn, Xt, mm, n, m, xs_row=zero, xt_row=zero, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
gwd_padded_l1_batch(Xs, nn, Xt, mm, n, m, xs_row=zero, xt_row=zero, out=out)
torch.cuda.synchronize()
print("batch of %d: %.2f ms (%.1f us per solve)" % (P, (time.perf_counter() - t0) * 1e3, (time.perf_counter() - t0) * 1e6 / P))
for _ in range(3):
    c = gwd_padded_l1(Xs, Xt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(36):
    c = gwd_padded_l1(Xs, Xt)
torch.cuda.synchronize()
print("36 single solves: %.2f ms (%.1f us per solve); equal %s" % ((time.perf_counter() - t0) * 1e3, (time.perf_counter() - t0) * 1e6 / 36,
                                                                   float(c) == float(out[0])))
