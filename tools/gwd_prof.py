import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from event_representation_study_amd.engine import gwd_padded_l1
rng = np.random.default_rng(77)
n, m = 12500, 14400
Xs = torch.from_numpy(rng.random((n, 4))).cuda()
Xt = torch.from_numpy(rng.random((m, 14)) * np.array([255.0] * 12 + [1.0, 1.0])).cuda()
for i in range(3):
    c = gwd_padded_l1(Xs, Xt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(36):
    c = gwd_padded_l1(Xs, Xt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("36 calls: issue %.2f ms, total %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), float(c))
costs = torch.zeros(36, dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(36):
    costs[i] = gwd_padded_l1(Xs + 1e-3 * i, Xt)
torch.cuda.synchronize(); t2 = time.perf_counter()
print("bench-style loop: %.2f ms" % ((t2 - t0) * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(36):
    x = Xs + 1e-3 * i
torch.cuda.synchronize(); t2 = time.perf_counter()
print("just the adds: %.2f ms" % ((t2 - t0) * 1e3))
c = gwd_padded_l1(Xs, Xt)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(36):
    costs[i] = c
torch.cuda.synchronize(); t2 = time.perf_counter()
print("just costs[i] = c: %.2f ms" % ((t2 - t0) * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(36):
    c = gwd_padded_l1(Xs + 1e-3 * i, Xt)
torch.cuda.synchronize(); t2 = time.perf_counter()
print("calls with fresh Xs: %.2f ms" % ((t2 - t0) * 1e3))
