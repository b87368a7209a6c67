"""CPU: the EST value MLP as an exact piecewise-linear table, and the oracle's restatement of
QuantizationLayer.forward, against outputs of the reference's own layer (tests/golden/make_golden_est.py)."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def est_golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "est.npz"))


def _weights(g):
    from event_representation_study_amd.est import mlp_weights
    return mlp_weights({k[2:]: g[k] for k in g.files if k.startswith("w_")})


def test_piecewise_linear_table_is_the_mlp(est_golden):
    from event_representation_study_amd.est import PiecewiseLinearKernel
    k = PiecewiseLinearKernel(_weights(est_golden))
    assert 8 < len(k) < 20000 and np.all(np.diff(k.edges) > 0)   # the trained kernel has ~100 kinks inside [-1, 1]
    rng = np.random.default_rng(0)
    u = np.concatenate([rng.uniform(-1, 1, 20000), k.edges, k.edges[1:-1] - 1e-12, k.edges[1:-1] + 1e-12,
                        np.linspace(-1, 1, 4001)])
    u = np.clip(u, -1, 1)
    truth = k.mlp(u)                                   # the MLP in float64
    scale = np.abs(truth).max()
    assert np.abs(k(u) - truth).max() <= 1e-10 * scale  # exact up to float64 rounding: no kink is missed
    # and against the reference's own float32 forward on a grid
    ref = est_golden["mlp_f"].astype(np.float64)
    assert np.abs(k(est_golden["mlp_u"].astype(np.float64)) - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    # bucket index: every bucket starts at or before the piece containing its left edge
    left = k.lo + (k.hi - k.lo) * np.arange(k.nbucket) / k.nbucket
    piece = np.clip(np.searchsorted(k.edges[1:-1], left, side="right"), 0, len(k) - 1)
    assert np.all(k.bucket <= piece) and np.all(piece - k.bucket <= 1)


def test_oracle_est_voxel_vs_reference(oracle, est_golden):
    g = est_golden
    dim = tuple(int(v) for v in g["dim"])
    got = oracle.est_voxel(g["events"], dim, _weights(g))
    want = g["voxel"]
    assert got.shape == want.shape and got.dtype == np.float32
    # float32 matmuls in a different order than torch's: compare at the scale of the grid
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    assert np.array_equal(got == 0, want == 0) or np.abs(got[(got == 0) != (want == 0)]).max() < 1e-6
