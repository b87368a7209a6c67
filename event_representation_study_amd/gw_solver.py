"""Entropic Gromov-Wasserstein on the matrix cores -- an EXTENSION.

Nothing the reference runs computes this: its live GWD is the closed form of ``compute_otmi.OTMI`` (row A9,
``engine.gwd_padded_l1``); its only true-GW call is dead code (``ot.gromov.gromov_wasserstein(Ks, Kt, p, q,
"kl_loss")``, representation_search/gromov_wasserstein.py:62-69).  This module is the solver BASELINE.json's
north_star sketches -- pairwise cost tensor + Sinkhorn projections, the ``h1(C1) T h2(C2)^T`` contraction as two
MFMA GEMMs -- restated from POT's published ``entropic_gromov_wasserstein`` with fixed iteration counts.  PARITY
UNPINNED against POT (absent); checked against ``oracle/gw_oracle.py``.  Its scores are NOT comparable with the
C_p values the reference publishes (those come from A9).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check

LOSSES = {"square_loss": 0, "kl_loss": 1}


def entropic_gromov_wasserstein(C1, C2, p=None, q=None, loss_fun="square_loss", epsilon=0.1, outer_iters=10,
                                sinkhorn_iters=100, precision="f64", return_plan=True, device=None):
    """C1 (n, n), C2 (m, m) structure matrices (array-likes or CUDA tensors), p, q marginals (default uniform).
    Returns (T, gw): the (n, m) float64 plan (CUDA tensor, or None when return_plan is False) and the 0-dim
    float64 loss, both left on the device (no host synchronisation)."""
    if not torch.cuda.is_available():
        raise _lib.EvrepError("no HIP device visible: the solver runs on an MI355X only (no CPU fallback)")
    lib = _lib.load()
    dev = torch.device(device) if device is not None else (
        C1.device if isinstance(C1, torch.Tensor) and C1.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    as_dev = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))).to(dev, torch.float64).contiguous()  # noqa: E731
    A, Bm = as_dev(C1), as_dev(C2)
    if A.dim() != 2 or Bm.dim() != 2 or A.shape[0] != A.shape[1] or Bm.shape[0] != Bm.shape[1]:
        raise ValueError("C1 and C2 must be square matrices")
    n, m = int(A.shape[0]), int(Bm.shape[0])
    pv = as_dev(np.full(n, 1.0 / n) if p is None else p)
    qv = as_dev(np.full(m, 1.0 / m) if q is None else q)
    if pv.numel() != n or qv.numel() != m:
        raise ValueError("p / q do not match C1 / C2")
    if loss_fun not in LOSSES:
        raise ValueError("loss_fun must be 'square_loss' or 'kl_loss'")
    prec = {"f64": _lib.F64, "f32": _lib.F32}[precision]
    scratch = torch.empty(int(lib.evrep_gw_scratch_bytes(n, m, prec)), dtype=torch.uint8, device=dev)
    T = torch.empty((n, m), dtype=torch.float64, device=dev) if return_plan else None
    gw = torch.empty((), dtype=torch.float64, device=dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)  # noqa: E731
    with torch.cuda.device(dev):
        check(lib.evrep_entropic_gw(ptr(A), n, ptr(Bm), m, ptr(pv), ptr(qv), LOSSES[loss_fun], float(epsilon),
                                    int(outer_iters), int(sinkhorn_iters), prec, ptr(scratch), ptr(T), ptr(gw),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "evrep_entropic_gw")
    return T, gw


def flops_per_outer_iteration(n, m):
    """The two GEMMs of the tensor product: h1(C1) T (2 n^2 m) and (.) h2(C2)^T (2 n m^2)."""
    return 2.0 * n * n * m + 2.0 * n * m * m
