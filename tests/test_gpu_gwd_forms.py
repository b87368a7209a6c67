"""The two forms of the GWD tile kernel against the oracle's float64 closed form (compute_otmi.py:61-93).

Clouds of up to 15 dimensions take the split form (each float32 coordinate as three bfloat16 terms on the bfloat16 matrix
pipe, r03); wider ones keep the float32 MFMA chain.  Both must hold the 1e-5 budget with a wide margin (1e-7 asserted here;
5e-9-class measured) for every instantiation, for ragged sizes around the 128-point tile edge, and for either cloud being
the larger one (the tiles behind the smaller cloud take the one-sided path).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from event_representation_study_amd import engine
    return engine


@pytest.mark.parametrize("ds,dt", [(4, 14), (3, 3), (14, 4), (10, 12), (15, 15), (1, 2),   # split form: 2 / 6 steps per cloud
                                   (4, 20), (20, 4), (16, 16), (32, 5)])                    # float32 form
@pytest.mark.parametrize("n,m", [(700, 900), (900, 700), (128, 129), (257, 256), (5, 300)])
def test_forms_against_oracle(eng, oracle, ds, dt, n, m):
    rng = np.random.default_rng(1000 * ds + 10 * dt + n)
    Xs = rng.random((n, ds))
    Xt = rng.random((m, dt)) * np.linspace(1.0, 255.0, dt)
    got = float(eng.gwd_padded_l1(Xs, Xt).item())
    ref = oracle.gwd(Xs, Xt)
    assert abs(got - ref) <= 1e-7 * ref, (got, ref)


def test_split_form_keeps_the_float32_forms_accuracy(eng, oracle):
    """The same clouds through both forms (a zero column widens the second cloud past the split form's limit without
    changing any distance): both within 2e-8 of the float64 value, and of each other."""
    rng = np.random.default_rng(99)
    n, m = 1500, 1700
    Xs = rng.random((n, 4))
    Xt = rng.random((m, 14)) * np.array([255.0] * 12 + [1.0, 1.0])
    wide = np.concatenate([Xt, np.zeros((m, 2))], axis=1)   # 16 dimensions: float32 form
    ref = oracle.gwd(Xs, Xt)
    split = float(eng.gwd_padded_l1(Xs, Xt).item())
    f32 = float(eng.gwd_padded_l1(Xs, wide).item())
    assert abs(split - ref) <= 2e-8 * ref, (split, ref)
    assert abs(f32 - ref) <= 2e-8 * ref, (f32, ref)


def test_identical_clouds_cost_exactly_zero_in_both_forms(eng):
    rng = np.random.default_rng(5)
    for d in (4, 14, 20):
        X = rng.random((333, d))
        assert float(eng.gwd_padded_l1(X, X).item()) == 0.0
