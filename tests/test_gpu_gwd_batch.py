"""Config 4 on the device (r03): the batched GWD entry point and the device harness of otmi().

* evrep_gwd_padded_l1_batch: costs equal the single-solve entry point BIT FOR BIT (same tile sums, same summation order),
  for ragged pairs, pairs sharing a cloud, and against the oracle's closed form within 1e-5.
* evrep_otmi_event_clouds / evrep_otmi_rep_clouds: the device point clouds equal the host ones
  (compute_otmi.otmi_point_clouds, itself pinned to the reference-derived oracle in tests/test_host_logic_cpu.py) bit for bit.
* otmi_batch: the reference's golden otmi() cost within 1e-5.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _clouds(rng, sizes, d, scale=1.0):
    return [rng.random((n, d)) * scale for n in sizes]


def test_batch_equals_single_solves_bit_for_bit(oracle):
    from event_representation_study_amd import engine as eng
    rng = np.random.default_rng(5)
    ns, ms = [700, 1500, 129, 2, 300], [900, 1300, 128, 260, 300]
    Xs, Xt = _clouds(rng, ns, 4), _clouds(rng, ms, 14, 255.0)
    dev = torch.device("cuda:0")
    n_cap, m_cap = 1600, 1400
    big_s = torch.zeros((len(ns) * n_cap, 4), dtype=torch.float64, device=dev)
    big_t = torch.zeros((len(ms) * m_cap, 14), dtype=torch.float64, device=dev)
    for p, (a, b) in enumerate(zip(Xs, Xt)):
        big_s[p * n_cap: p * n_cap + len(a)] = torch.from_numpy(a).to(dev)
        big_t[p * m_cap: p * m_cap + len(b)] = torch.from_numpy(b).to(dev)
    n = torch.tensor(ns, dtype=torch.int64, device=dev)
    m = torch.tensor(ms, dtype=torch.int64, device=dev)
    costs = eng.gwd_padded_l1_batch(big_s, n, big_t, m, n_cap, m_cap).cpu().numpy()
    single = np.array([float(eng.gwd_padded_l1(a, b).item()) for a, b in zip(Xs, Xt)])
    assert np.array_equal(costs, single), (costs, single)
    for p in (0, 2):
        ref = oracle.gwd(Xs[p], Xt[p])
        assert abs(costs[p] - ref) <= 1e-5 * ref
    # explicit row tables: every pair reads the SAME source cloud, target clouds in reverse order
    xs_row = torch.zeros(len(ns), dtype=torch.int64, device=dev)
    xt_row = torch.arange(len(ms) - 1, -1, -1, dtype=torch.int64, device=dev) * m_cap
    n0 = torch.full((len(ns),), ns[0], dtype=torch.int64, device=dev)
    c2 = eng.gwd_padded_l1_batch(big_s, n0, big_t, m.flip(0).contiguous(), n_cap, m_cap, xs_row=xs_row, xt_row=xt_row).cpu().numpy()
    s2 = np.array([float(eng.gwd_padded_l1(Xs[0], b).item()) for b in Xt[::-1]])
    assert np.array_equal(c2, s2)
    # an empty and an oversized cloud cost NaN, the other pairs are untouched
    n_bad = n.clone(); n_bad[1] = 0; n_bad[3] = n_cap + 1
    c3 = eng.gwd_padded_l1_batch(big_s, n_bad, big_t, m, n_cap, m_cap).cpu().numpy()
    assert np.isnan(c3[1]) and np.isnan(c3[3]) and np.array_equal(c3[[0, 2, 4]], single[[0, 2, 4]])


@pytest.mark.parametrize("sizes,S", [((6000, 6701, 7402), 240),
                                     # windows longer / shorter than slices x threads of the sliced harness kernels, an odd cut
                                     ((50000, 1500, 33001), 239)])
def test_device_point_clouds_equal_the_host_harness_bit_for_bit(sizes, S):
    from event_representation_study_amd import engine as eng
    from event_representation_study_amd.representations.representation_search.compute_otmi import otmi_point_clouds
    from event_representation_study_amd.synthetic import make_events
    H, W, C = 240, 304, 5
    rng = np.random.default_rng(11)
    wins, reps = [], []
    for b in range(3):
        ev = make_events(sizes[b], W, H, seed=40 + b, polarity="pm1" if b != 1 else "01")
        if b == 2:                      # a skewed window: most events in the last quadrant, which is then the skipped one
            ev[: len(ev) // 2, 0] = rng.integers(W // 2 + 3, W, len(ev) // 2)
            ev[: len(ev) // 2, 1] = rng.integers(H // 2 + 3, H, len(ev) // 2)
        wins.append(ev)
    for r in range(2):
        for b in range(3):
            rep = rng.random((S, S, C)) * (rng.random((S, S, 1)) < 0.3)      # 70 % of the pixels are all-zero rows
            rep[:8] = 114.0                                                   # letterbox rows count as points
            reps.append(rep)
    dev = torch.device("cuda:0")
    offs = np.zeros(len(wins) + 1, dtype=np.int64)
    np.cumsum([len(w) for w in wins], out=offs[1:])
    events = torch.from_numpy(np.concatenate(wins)).to(dev)
    Xs, n, quad = eng.otmi_event_clouds(events, torch.from_numpy(offs), H, W)
    rep_t = torch.from_numpy(np.stack(reps)).to(dev)
    Xt, m, m_cap = eng.otmi_rep_clouds(rep_t, quad, len(wins))
    Xs, n, Xt, m = Xs.cpu().numpy(), n.cpu().numpy(), Xt.cpu().numpy(), m.cpu().numpy()
    for r in range(2):
        for b in range(3):
            host = otmi_point_clouds(torch.from_numpy(wins[b]), reps[r * 3 + b], H, W, S)
            assert len(host) == 3
            for k, (hs, ht) in enumerate(host):
                assert n[b, k] == len(hs) and m[r * 3 + b, k] == len(ht)
                assert np.array_equal(Xs[b, k, : len(hs)], hs.astype(np.float64))
                assert np.array_equal(Xt[r * 3 + b, k, : len(ht)], ht)


def test_otmi_batch_against_the_reference_golden():
    from event_representation_study_amd import engine as eng
    from event_representation_study_amd.representations.representation_search.compute_otmi import otmi, otmi_batch
    g = load_golden("gwd")
    ev = np.ascontiguousarray(g["otmi_events"], dtype=np.int32)
    H, W, S = int(g["otmi_H"]), int(g["otmi_W"]), int(g["otmi_S"])
    rep = np.asarray(g["otmi_rep"], dtype=np.float64)
    dev = torch.device("cuda:0")
    offs = torch.tensor([0, len(ev), 2 * len(ev)], dtype=torch.int64)
    events = torch.from_numpy(np.concatenate([ev, ev])).to(dev)
    reps = torch.from_numpy(np.stack([rep, rep])[None]).to(dev)            # (R = 1, B = 2, S, S, C)
    mean, quads = eng.otmi_batch(events, offs, reps, H, W)
    ref = float(g["otmi_cost"])
    assert abs(float(mean[0, 0].item()) - ref) <= 1e-5 * ref
    assert torch.equal(quads[0, 0], quads[0, 1])
    # the per-sample mirror (host harness + single solves) and the batched mirror agree to rounding of the mean
    per_sample = otmi(torch.from_numpy(ev.copy()), rep, H, W, S)
    assert abs(per_sample - float(mean[0, 0].item())) <= 1e-12 * ref
    costs = otmi_batch([torch.from_numpy(ev.copy())] * 2, [rep, rep], H, W, S)
    assert abs(costs[0] - per_sample) <= 1e-12 * ref and costs[0] == costs[1]


def test_otmi_routes_agree_bit_for_bit():
    """otmi() takes the device harness for integer events; the value equals the host route's (the reference's own structure:
    host quadrant bookkeeping + one solve per quadrant) bit for bit, for both polarity encodings and a skewed window; inputs the
    device route does not cover fall back to the host route."""
    from event_representation_study_amd.representations.representation_search import compute_otmi as co
    from event_representation_study_amd.synthetic import make_events
    H, W, S = 240, 304, 240
    rng = np.random.default_rng(21)
    for b, enc in enumerate(("pm1", "01", "pm1")):
        ev = make_events(9000 + 500 * b, W, H, seed=70 + b, polarity=enc)
        if b == 2:
            ev[: len(ev) // 2, 0] = rng.integers(W // 2 + 3, W, len(ev) // 2)
        rep = rng.random((S, S, 5)) * (rng.random((S, S, 1)) < 0.35)
        a = co.otmi(torch.from_numpy(ev.copy()), rep, H, W, S)
        h = co._otmi_host(torch.from_numpy(ev.copy()), rep, H, W, S)
        assert a == h, (a, h)
        assert co.otmi(ev.astype(np.float64), rep, H, W, S) == h            # float events: host route, same value
    # an empty quadrant: the reference's min() of nothing raises; the device route's NaN must not hide that
    ev = make_events(3000, W, H, seed=99)
    ev = ev[(ev[:, 0] <= W // 2 - 1) | (ev[:, 1] <= H // 2 - 1)]            # nothing in the bottom-right quadrant
    rep = rng.random((S, S, 5))
    with pytest.raises(ValueError):
        co.otmi(torch.from_numpy(ev.copy()), rep, H, W, S)

