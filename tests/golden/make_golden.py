#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (``/root/reference`` does not exist on the GPU
box): ``python tests/golden/make_golden.py``.  Nothing from the reference is
copied: its modules are imported from where they lie and executed on seeded
synthetic streams; only inputs and outputs (data) are written.

Third-party packages the reference needs but this image lacks are replaced by
the smallest possible stand-ins *in sys.modules of this process only*
(SURVEY.md section 8(c)):

* ``numba``          -> identity ``jit`` decorator; the loop body that runs is the
                        reference's own arithmetic (time_surface.py:66-74).
* ``torch_scatter``  -> ``scatter`` restated on torch.scatter_add_/scatter_reduce_
                        following torch_scatter's documented semantics (sum; mean =
                        sum / clamp(count, 1); max with empty -> 0).  PARITY
                        UNPINNED at this one call boundary (operations.py:17,20,29).
* ``tonic``          -> empty module (only the in-repo branches of the dispatcher
                        are exercised).
* ``ot`` (POT)       -> ``unif`` and ``gromov.sampled_gromov_wasserstein`` restated
                        from POT >= 0.8's published algorithm for ``max_iter=0``:
                        T = p q^T, ``gw_dist_estimated`` = mean over 2 stacked
                        evaluations of ``loss_fun``.  PARITY UNPINNED for the
                        returned scalar (compute_otmi.py:81-93).
"""
# What the two arithmetic stand-ins restate, so the claim can be audited against the upstream sources (both packages
# are absent from this image and un-vendored by the reference; file names as published, quoted from memory):
#
# torch_scatter (csrc/cpu/scatter_cpu.cpp, `scatter_cpu`; python wrapper torch_scatter/scatter.py):
#   * the CPU kernel is ONE sequential loop over `src` in index order per output slice -- `for i in range(E): out[idx] =
#     reduce(out[idx], src[i])` -- hence "events of a pixel are added in time order";
#   * reduce="sum": out starts at 0 (`torch.zeros`), `+=`;
#   * reduce="mean": `scatter_sum`, then `count = scatter_sum(ones)`, `count[count < 1] = 1` (`count.clamp_(1)`),
#     `out.true_divide_(count)`  (scatter.py `scatter_mean`);
#   * reduce="max": `out.fill_(std::numeric_limits<scalar_t>::lowest())`, update where `src > out`, and afterwards
#     `out.masked_fill_(arg_out == index.size(dim), 0)` -- slots nothing was scattered into become 0.
#   The stand-in below does the same with torch.scatter_add_ / scatter_reduce_("amax") + the masked fill; for sums it
#   relies on torch's CPU scatter_add_ being the same sequential in-order loop (it is: index_put-free, single thread
#   here because torch.set_num_threads(1)).
#
# POT (ot/gromov/_estimators.py, `sampled_gromov_wasserstein` and `GW_distance_estimation`):
#   * `T = np.outer(p, q)`; the `for cpt in range(max_iter)` body never runs for max_iter = 0;
#   * with log=True: `log['gw_dist_estimated'], log['gw_dist_std'] = GW_distance_estimation(C1, C2, loss_fun, p, q, T,
#     nb_samples_p, nb_samples_q, std=True)`, which draws index samples, evaluates `loss_fun` on the sampled
#     sub-matrices `nb_samples_q` times, stacks the results and returns their `mean()` (and std);
#   * the reference's loss_fun IGNORES its arguments and returns the full padded |Ks - Kt| (compute_otmi.py:73-75), so
#     every one of the stacked evaluations is that same matrix whatever was sampled: the mean over the stack is its
#     plain mean, deterministic despite POT's unseeded sampling.  The stand-in stacks two evaluations.
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from event_representation_study_amd.synthetic import make_events, to_structured  # noqa: E402

torch.set_num_threads(1)


# --------------------------------------------------------------------------- stand-ins
def _install_standins():
    numba = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    numba.jit = jit
    numba.njit = jit
    sys.modules["numba"] = numba

    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
        assert src.dim() == 1 and index.dim() == 1 and out is None
        if reduce in ("sum", "add"):
            return torch.zeros(dim_size, dtype=src.dtype).scatter_add_(0, index, src)
        if reduce == "mean":
            s = torch.zeros(dim_size, dtype=src.dtype).scatter_add_(0, index, src)
            c = torch.zeros(dim_size, dtype=src.dtype).scatter_add_(0, index, torch.ones_like(src))
            c[c < 1] = 1
            return s.true_divide_(c)
        if reduce == "max":
            lowest = torch.finfo(src.dtype).min
            o = torch.full((dim_size,), lowest, dtype=src.dtype)
            o.scatter_reduce_(0, index, src, reduce="amax", include_self=True)
            o.masked_fill_(o == lowest, 0)
            return o
        raise ValueError(reduce)

    ts.scatter = scatter
    sys.modules["torch_scatter"] = ts

    tonic = types.ModuleType("tonic")
    tonic.transforms = types.ModuleType("tonic.transforms")
    sys.modules["tonic"] = tonic
    sys.modules["tonic.transforms"] = tonic.transforms

    ot = types.ModuleType("ot")
    ot.gromov = types.ModuleType("ot.gromov")

    def unif(n):
        return np.ones((n,)) / n

    def sampled_gromov_wasserstein(C1, C2, p, q, loss_fun, nb_samples_grad=100, epsilon=1,
                                   max_iter=500, log=False, verbose=False, random_state=None):
        assert max_iter == 0, "stand-in only covers the reference's call (max_iter=0)"
        T = np.outer(p, q)
        # GW_distance_estimation with std=True: nb_samples_q = 2 stacked loss evaluations.
        vals = np.stack([loss_fun(C1, C2) for _ in range(2)], axis=2)
        if log:
            return T, {"gw_dist_estimated": np.mean(vals),
                       "gw_dist_std": float(np.sum(np.std(vals, axis=2) ** 2) ** 0.5)}
        return T

    ot.unif = unif
    ot.gromov.sampled_gromov_wasserstein = sampled_gromov_wasserstein
    sys.modules["ot"] = ot
    sys.modules["ot.gromov"] = ot.gromov


def _import_reference():
    _install_standins()
    os.chdir(REF)  # optimized_representation.py:4-9 builds sys.path from os.getcwd()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "representations"))
    from representations import gen1_transforms, gen4_transforms  # noqa
    from representations.event_stack import EventStack
    from representations.time_surface import ToTimesurface
    from representations.tore import events2ToreFeature
    from representations.optimized_representation import get_optimized_representation
    from representations.representation_search.mixed_density_event_stack import MixedDensityEventStack
    from representations.representation_search import compute_otmi
    from representations.representation_search.gromov_wasserstein import compute_repr
    return dict(gen1=gen1_transforms, gen4=gen4_transforms, EventStack=EventStack,
                ToTimesurface=ToTimesurface, tore=events2ToreFeature,
                opt=get_optimized_representation, MDES=MixedDensityEventStack,
                otmi_mod=compute_otmi, compute_repr=compute_repr)


# --------------------------------------------------------------------------- cases
CASES = [
    # name, W, H, N, seed, kwargs
    ("c1_304x240_n10000_pm1", 304, 240, 10000, 101, dict(polarity="pm1")),
    ("c1_304x240_n10000_01", 304, 240, 10000, 102, dict(polarity="01")),
    ("dense_64x48_n61440_pm1", 64, 48, 61440, 103, dict(polarity="pm1")),
    ("s_80x60_n4095_pm1", 80, 60, 4095, 104, dict(polarity="pm1")),
    ("s_80x60_n4097_01", 80, 60, 4097, 105, dict(polarity="01")),
    ("s_80x60_n5000_pm1", 80, 60, 5000, 106, dict(polarity="pm1")),      # N mod 3 == 2
    ("s_80x60_n5001_pm1", 80, 60, 5001, 107, dict(polarity="pm1")),      # N mod 3 == 0
    ("s_80x60_n5002_01", 80, 60, 5002, 108, dict(polarity="01")),        # N mod 3 == 1
    ("duplast_80x60_n3000_pm1", 80, 60, 3000, 109, dict(polarity="pm1", dup_last=7)),
    ("shortspan_40x30_n2000_pm1", 40, 30, 2000, 110, dict(polarity="pm1", span_us=5)),  # equal idx -> dead TS slices
    ("single_pos_80x60_n2000", 80, 60, 2000, 111, dict(polarity="pm1", single_polarity=1)),
    ("single_neg_80x60_n2000", 80, 60, 2000, 112, dict(polarity="pm1", single_polarity=-1)),
    ("single_zero_80x60_n2000", 80, 60, 2000, 113, dict(polarity="01", single_polarity=0)),
    ("tiny_16x12_n13_pm1", 16, 12, 13, 114, dict(polarity="pm1")),
    ("tiny_16x12_n2_01", 16, 12, 2, 115, dict(polarity="01")),
]
BIG = ("big_640x480_n50000_pm1", 640, 480, 50000, 201, dict(polarity="pm1"))

ALL_FUNCS = ["timestamp", "polarity", "count", "timestamp_pos", "timestamp_neg", "count_pos", "count_neg"]
ALL_AGGS = ["sum", "mean", "max", "variance"]


def ts_like_dispatch(ref, rec, H, W):
    """TimeSurface exactly as gen1_transforms.py:69-85 drives it (without the *255)."""
    rec = rec.copy()
    rec["p"] = ((rec["p"] + 1) / 2).astype(np.int8)
    tr = ref["ToTimesurface"](sensor_size=(W, H, 2), surface_dimensions=None, tau=50000, decay="exp")
    t = rec["t"]
    t_norm = (t - t[0]) / (t[-1] - t[0]) * 6
    idx = np.searchsorted(t_norm, np.arange(6) + 1)
    rep = tr(rec, idx)
    rep = rep.reshape((-1, rep.shape[-2], rep.shape[-1])).transpose(1, 2, 0)
    return np.ascontiguousarray(rep), idx


def es_like_dispatch(ref, rec, H, W):
    rec = rec.copy()
    rec["p"] = (rec["p"] + 1) // 2
    tr = ref["EventStack"](12, rec.shape[0], H, W)
    pre = tr.pre_stack(rec, rec[-1]["t"])
    post = tr.post_stack(pre)
    return np.ascontiguousarray(post.transpose(0, 1, 3, 2)[..., 0])


def tore_like_dispatch(ref, rec):
    x, y, ts, pol = rec["x"], rec["y"], rec["t"], rec["p"]
    x = x - min(x) + 1
    y = y - min(y) + 1
    frame = (max(y), max(x))
    return ref["tore"](x, y, ts, pol, ts[-1], 6, frame)


def voxel5(ref, ev, W, H, bins=5):
    t = ev[:, 2].astype(np.float64)
    t = (t - t[0]) / (t[-1] - t[0])           # gromov_wasserstein.py:96
    return ref["compute_repr"](ev[:, 0], ev[:, 1], t, ev[:, 3], W, H, bins=bins)


def run_case(ref, name, W, H, N, seed, kw, full=True):
    ev = make_events(N, W, H, seed=seed, **kw)
    rec = to_structured(ev)
    out = {"events": ev, "W": W, "H": H}
    with np.errstate(all="ignore"):
        out["ergo12"] = ref["opt"](rec.copy(), N, H, W)
        out["event_stack"] = es_like_dispatch(ref, rec, H, W)
        if N >= 2 and ev[-1, 2] != ev[0, 2]:
            out["time_surface"], out["ts_idx"] = ts_like_dispatch(ref, rec, H, W)
        out["tore"] = tore_like_dispatch(ref, rec)
        if N >= 2 and ev[-1, 2] != ev[0, 2]:
            out["voxel5"] = voxel5(ref, ev, W, H)
    return out


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def main():
    ref = _import_reference()
    os.makedirs(HERE, exist_ok=True)
    manifest = {"numpy": np.__version__, "torch": torch.__version__, "cases": []}

    for (name, W, H, N, seed, kw) in CASES:
        out = run_case(ref, name, W, H, N, seed, kw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        manifest["cases"].append(dict(name=name, W=W, H=H, N=N, seed=seed, kw=kw,
                                      keys=sorted(k for k in out if k not in ("events", "W", "H"))))
        print("wrote", name, {k: getattr(v, "shape", v) for k, v in out.items()})

    # ---- every (window, func, agg) triple on a small frame, both encodings ---------------
    for enc, seed in (("pm1", 301), ("01", 302)):
        W, H, N = 40, 30, 3001
        ev = make_events(N, W, H, seed=seed, polarity=enc)
        triples = [(w, f, a) for w in range(7) for f in ALL_FUNCS for a in ALL_AGGS]
        wi = [t[0] for t in triples]
        fu = [t[1] for t in triples]
        ag = [t[2] for t in triples]
        m = ref["MDES"](len(triples), N, H, W, (wi, fu, ag), "SBN")
        rep = m.stack(to_structured(ev))
        name = "mdes_all_triples_40x30_n3001_" + enc
        np.savez_compressed(os.path.join(HERE, name + ".npz"), events=ev, W=W, H=H, rep=rep,
                            windows=np.array(wi), funcs=np.array(fu), aggs=np.array(ag))
        manifest["cases"].append(dict(name=name, W=W, H=H, N=N, seed=seed, kw=dict(polarity=enc), keys=["rep"]))
        print("wrote", name, rep.shape)

    # ---- MDES failure semantics: window None -> zero channel (optimization.py:48-52) -------
    W, H, N = 40, 30, 999
    ev = make_events(N, W, H, seed=303)
    m = ref["MDES"](4, N, H, W, ([0, None, 6, None], ["count", None, "polarity", None], ["sum", None, "sum", None]), "SBN")
    rep = m.stack(to_structured(ev))
    np.savez_compressed(os.path.join(HERE, "mdes_none_channels_40x30_n999.npz"), events=ev, W=W, H=H, rep=rep)
    manifest["cases"].append(dict(name="mdes_none_channels_40x30_n999", W=W, H=H, N=N, seed=303, kw={}, keys=["rep"]))

    # ---- dispatcher (A1): in-repo branches, *255 and in-place mutation of ["p"] ------------
    W, H, N = 80, 60, 5000
    for enc, seed in (("pm1", 401), ("01", 402)):
        ev = make_events(N, W, H, seed=seed, polarity=enc)
        d = {"events": ev, "W": W, "H": H}
        for label, rname in (("mdes", "MixedDensityEventStack"), ("event_stack", "EventStack"),
                             ("tore", "<function events2ToreFeature at 0x0>"), ("time_surface", "ToTimesurface")):
            rec = to_structured(ev)
            with np.errstate(all="ignore"):
                rep = ref["gen1"].get_item_transform(rec, rname, None, H, W, N, 50000)
                rep4 = ref["gen4"].get_item_transform(to_structured(ev), rname, None, H, W, N)
            assert np.array_equal(rep, rep4, equal_nan=True)
            d["rep_" + label] = rep
            d["p_after_" + label] = rec["p"].copy()
        name = "dispatch_80x60_n5000_" + enc
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        manifest["cases"].append(dict(name=name, W=W, H=H, N=N, seed=seed, kw=dict(polarity=enc),
                                      keys=sorted(k for k in d if k not in ("events", "W", "H"))))
        print("wrote", name)

    # ---- full-size digest (640x480, N=50000) ---------------------------------------------
    name, W, H, N, seed, kw = BIG
    out = run_case(ref, name, W, H, N, seed, kw)
    rng = np.random.default_rng(7)
    big = {"W": W, "H": H, "N": N, "seed": seed}
    for k in ("ergo12", "event_stack", "time_surface", "tore", "voxel5"):
        a = np.ascontiguousarray(out[k])
        flat = a.reshape(-1)
        pos = rng.integers(0, flat.size, size=8192)
        nz = np.flatnonzero(flat != (flat[0] if k in ("time_surface", "tore") else 0))
        if nz.size:
            pos[:4096] = nz[rng.integers(0, nz.size, size=4096)]
        big[k + "_shape"] = np.array(a.shape)
        big[k + "_sha256"] = digest(a)
        big[k + "_pos"] = pos
        big[k + "_val"] = flat[pos]
        big[k + "_sum"] = np.float64(flat.astype(np.float64).sum())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **big)
    manifest["cases"].append(dict(name=name, W=W, H=H, N=N, seed=seed, kw=kw, keys=["digest"]))
    print("wrote", name)

    # ---- GWD (A9/A10) --------------------------------------------------------------------
    cm = ref["otmi_mod"]
    rng = np.random.default_rng(11)
    g = {}
    # A9: OTMI on raw point clouds. Xs float32 (torch int / int -> float32), Xt float64.
    for tag, n, m_, dt in (("a", 300, 257, 14), ("b", 128, 400, 4), ("c", 513, 513, 8)):
        Xs = rng.random((n, 4)).astype(np.float32)
        Xt = rng.random((m_, dt)) * np.array([255.0] * (dt - 2) + [1.0, 1.0])
        o = cm.OTMI(Xs, Xt, h=0.7, reg=0.05)
        T, cost = o.solve()
        g[f"{tag}_Xs"], g[f"{tag}_Xt"] = Xs, Xt
        g[f"{tag}_cost"] = np.float64(cost)
        g[f"{tag}_Ks_mean"] = np.float64(o.Ks.astype(np.float64).mean())
        g[f"{tag}_Kt_mean"] = np.float64(o.Kt.mean())
        g[f"{tag}_Ks_probe"] = o.Ks[:8, :8].astype(np.float64)
        g[f"{tag}_Kt_probe"] = o.Kt[:8, :8]
        g[f"{tag}_T00"] = np.float64(T[0, 0])
    # A10: the quadrant harness, events as a torch int tensor, rep letterboxed-like (S,S,C).
    W, H, S, N = 304, 240, 64, 6000
    ev = make_events(N, W, H, seed=501)
    repq = np.zeros((S, S, 3))
    m0 = rng.random((S, S)) < 0.15
    repq[m0] = rng.random((int(m0.sum()), 3)) * 255
    repq[:6] = 114.0
    repq[-6:] = 114.0
    cost = cm.otmi(torch.from_numpy(ev.copy()), repq, H, W, S)
    g["otmi_events"], g["otmi_rep"] = ev, repq
    g["otmi_H"], g["otmi_W"], g["otmi_S"] = H, W, S
    g["otmi_cost"] = np.float64(cost)
    np.savez_compressed(os.path.join(HERE, "gwd.npz"), **g)
    manifest["cases"].append(dict(name="gwd", keys=sorted(g)))
    print("wrote gwd", {k: float(v) for k, v in g.items() if k.endswith("cost")})

    # ---- F3: ev-licious events_to_voxel_grid (numpy variant), loaded from its file with stand-ins ----
    import importlib.util
    evl = types.ModuleType("evlicious")
    evl.Events = object                       # utils.py only uses the name in annotations
    sys.modules["evlicious"] = evl
    spec = importlib.util.spec_from_file_location(
        "evl_utils", os.path.join(REF, "ev-licious", "src", "evlicious", "tools", "utils.py"))
    evl_utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(evl_utils)

    class _Events:                            # the fields Events carries (io/utils/events.py:8-20), p in {-1,+1}
        def __len__(self):
            return len(self.x)

    v = {}
    for tag, W, H, N, seed in (("a", 80, 60, 6000, 601), ("b", 304, 240, 20000, 602)):
        ev = make_events(N, W, H, seed=seed)
        e = _Events()
        e.x, e.y = ev[:, 0].astype(np.uint16), ev[:, 1].astype(np.uint16)
        e.t, e.p = ev[:, 2].astype(np.int64), ev[:, 3].astype(np.int8)
        e.width, e.height = W, H
        v[tag + "_events"], v[tag + "_W"], v[tag + "_H"] = ev, W, H
        v[tag + "_raw5"] = evl_utils.events_to_voxel_grid(e, 5, normalize=False)
        v[tag + "_norm5"] = evl_utils.events_to_voxel_grid(e, 5, normalize=True)
        v[tag + "_raw12"] = evl_utils.events_to_voxel_grid(e, 12, normalize=False)
    np.savez_compressed(os.path.join(HERE, "evlicious_voxel.npz"), **v)
    manifest["cases"].append(dict(name="evlicious_voxel", keys=sorted(v)))
    print("wrote evlicious_voxel")

    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, default=str)


if __name__ == "__main__":
    main()
