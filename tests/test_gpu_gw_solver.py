"""-m gpu: the entropic Gromov-Wasserstein extension against oracle/gw_oracle.py.

Tolerances: float64 MFMA path 1e-9 relative on the plan and the loss (same recurrences, different summation
order); float32 MFMA path 1e-5 on the loss at epsilon chosen so that the float32 rounding of the tensor product
(~1e-6) is not amplified past the budget by the 1/epsilon of the Gibbs kernel.  PARITY UNPINNED against POT."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gauss_kernel(X, h=0.7):
    D2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    sig2 = D2.mean() / 2.0
    return np.exp(-D2 / (2.0 * h * h * sig2))


def _problem(n, m, seed):
    rng = np.random.default_rng(seed)
    C1 = _gauss_kernel(rng.random((n, 4)))                 # the reference's Ks / Kt are Gaussian kernels in (0, 1]
    C2 = _gauss_kernel(rng.random((m, 6)) * np.array([3.0, 3, 3, 3, 1, 1]))
    p = rng.random(n) + 0.5
    q = rng.random(m) + 0.5
    return C1, C2, p / p.sum(), q / q.sum()


@pytest.mark.parametrize("loss", ["square_loss", "kl_loss"])
@pytest.mark.parametrize("n,m", [(500, 437), (130, 257), (64, 64)])
def test_entropic_gw_f64_matches_oracle(loss, n, m):
    from oracle import gw_oracle
    from event_representation_study_amd.gw_solver import entropic_gromov_wasserstein
    C1, C2, p, q = _problem(n, m, seed=n + m)
    Tref, gref = gw_oracle.entropic_gromov_wasserstein(C1, C2, p, q, loss, epsilon=0.05, outer_iters=6, sinkhorn_iters=60)
    T, gw = entropic_gromov_wasserstein(C1, C2, p, q, loss, epsilon=0.05, outer_iters=6, sinkhorn_iters=60, precision="f64")
    T = T.cpu().numpy()
    assert abs(float(gw) - gref) <= 1e-9 * abs(gref), (float(gw), gref)
    np.testing.assert_allclose(T, Tref, rtol=1e-8, atol=1e-14)
    # marginals of a Sinkhorn plan: rows exact after the last row scaling
    np.testing.assert_allclose(T.sum(1), p, rtol=1e-10)


def test_entropic_gw_f32_within_budget():
    from oracle import gw_oracle
    from event_representation_study_amd.gw_solver import entropic_gromov_wasserstein
    n, m = 500, 437
    C1, C2, p, q = _problem(n, m, seed=7)
    Tref, gref = gw_oracle.entropic_gromov_wasserstein(C1, C2, p, q, "square_loss", epsilon=0.5, outer_iters=5, sinkhorn_iters=40)
    T, gw = entropic_gromov_wasserstein(C1, C2, p, q, "square_loss", epsilon=0.5, outer_iters=5, sinkhorn_iters=40, precision="f32")
    assert abs(float(gw) - gref) <= 1e-5 * abs(gref), (float(gw), gref)
    np.testing.assert_allclose(T.cpu().numpy(), Tref, rtol=2e-4, atol=1e-12)


def test_entropic_gw_identity_and_determinism():
    from event_representation_study_amd.gw_solver import entropic_gromov_wasserstein
    C1, C2, p, q = _problem(300, 300, seed=11)
    T1, g1 = entropic_gromov_wasserstein(C1, C1, p, p, "square_loss", epsilon=0.05, outer_iters=4, sinkhorn_iters=50)
    T2, g2 = entropic_gromov_wasserstein(C1, C1, p, p, "square_loss", epsilon=0.05, outer_iters=4, sinkhorn_iters=50)
    assert torch.equal(T1, T2) and float(g1) == float(g2)          # no atomics anywhere: bit-identical reruns
    # relabelling the points of the second space permutes the plan's columns and leaves the loss alone
    T3, g3 = entropic_gromov_wasserstein(C1, C2, p, q, "square_loss", epsilon=0.05, outer_iters=4, sinkhorn_iters=50)
    perm = np.random.default_rng(5).permutation(300)
    T4, g4 = entropic_gromov_wasserstein(C1, C2[np.ix_(perm, perm)], p, q[perm], "square_loss", epsilon=0.05,
                                         outer_iters=4, sinkhorn_iters=50)
    assert abs(float(g3) - float(g4)) <= 1e-10 * abs(float(g3))
    np.testing.assert_allclose(T4.cpu().numpy(), T3.cpu().numpy()[:, perm], rtol=1e-8, atol=1e-15)


def test_entropic_gw_odd_sizes_and_arguments():
    from oracle import gw_oracle
    from event_representation_study_amd.gw_solver import entropic_gromov_wasserstein
    C1, C2, p, q = _problem(17, 129, seed=3)                        # not multiples of any tile edge
    Tref, gref = gw_oracle.entropic_gromov_wasserstein(C1, C2, p, q, "kl_loss", 0.1, 3, 30)
    T, gw = entropic_gromov_wasserstein(C1, C2, p, q, "kl_loss", 0.1, 3, 30)
    assert abs(float(gw) - gref) <= 1e-9 * abs(gref)
    np.testing.assert_allclose(T.cpu().numpy(), Tref, rtol=1e-8, atol=1e-14)
    with pytest.raises(ValueError):
        entropic_gromov_wasserstein(C1, C2[:, :5], p, q)
    with pytest.raises(ValueError):
        entropic_gromov_wasserstein(C1, C2, p, q, loss_fun="l1")
