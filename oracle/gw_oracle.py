"""Float64 numpy restatement of POT's entropic Gromov-Wasserstein (PGD solver) -- TEST INFRASTRUCTURE ONLY.

SURVEY.md 8 row F5: the only true-GW call of the reference is the dead-code
``ot.gromov.gromov_wasserstein(Ks, Kt, p, q, "kl_loss")`` (representations/representation_search/
gromov_wasserstein.py:62-69); POT is absent from /root/reference and unpinned, so this restates POT's PUBLISHED
algorithm (ot/gromov/_utils.py: init_matrix, tensor_product, gwloss, gwggrad; ot/gromov/_bregman.py:
entropic_gromov_wasserstein; ot/bregman/_sinkhorn.py: sinkhorn_knopp) -- PARITY UNPINNED against POT itself.

One deliberate difference, shared with the HIP path: POT stops Sinkhorn on a marginal-error test every 10
iterations and the outer loop on ||T - Tprev||; both loops run a FIXED number of iterations here, so that the two
implementations execute the same recurrences and can be compared entry by entry.
"""
import numpy as np


def init_matrix(C1, C2, p, q, loss_fun="square_loss"):
    if loss_fun == "square_loss":
        f1, f2, h1, h2 = (lambda a: a ** 2), (lambda b: b ** 2), (lambda a: a), (lambda b: 2 * b)
    elif loss_fun == "kl_loss":
        f1, f2 = (lambda a: a * np.log(a + 1e-15) - a), (lambda b: b)
        h1, h2 = (lambda a: a), (lambda b: np.log(b + 1e-15))
    else:
        raise ValueError(loss_fun)
    constC1 = np.dot(np.dot(f1(C1), p.reshape(-1, 1)), np.ones((1, len(q))))
    constC2 = np.dot(np.ones((len(p), 1)), np.dot(q.reshape(1, -1), f2(C2).T))
    return constC1 + constC2, h1(C1), h2(C2)


def tensor_product(constC, hC1, hC2, T):
    return constC - np.dot(np.dot(hC1, T), hC2.T)


def gwloss(constC, hC1, hC2, T):
    return float(np.sum(tensor_product(constC, hC1, hC2, T) * T))


def sinkhorn_knopp(a, b, M, reg, iters):
    u = np.ones(len(a)) / len(a)
    v = np.ones(len(b)) / len(b)
    K = np.exp(M / (-reg))
    Kp = (1.0 / a).reshape(-1, 1) * K
    for _ in range(iters):
        v = b / np.dot(K.T, u)
        u = 1.0 / np.dot(Kp, v)
    return u.reshape(-1, 1) * K * v.reshape(1, -1)


def entropic_gromov_wasserstein(C1, C2, p, q, loss_fun="square_loss", epsilon=0.1, outer_iters=10, sinkhorn_iters=100):
    """-> (T, gw_dist).  T = p q^T; repeat: T = sinkhorn(p, q, gwggrad(T) = 2 tens(T), epsilon); gw = gwloss(T)."""
    C1, C2, p, q = (np.asarray(x, dtype=np.float64) for x in (C1, C2, p, q))
    T = np.outer(p, q)
    constC, hC1, hC2 = init_matrix(C1, C2, p, q, loss_fun)
    for _ in range(outer_iters):
        T = sinkhorn_knopp(p, q, 2.0 * tensor_product(constC, hC1, hC2, T), epsilon, sinkhorn_iters)
    return T, gwloss(constC, hC1, hC2, T)
