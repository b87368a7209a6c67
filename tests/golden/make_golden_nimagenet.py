#!/usr/bin/env python3
"""Golden vectors for row F4 (n_imagenet's reshape_then_* accumulators).

Run HERE (the container that has /root/reference); the fixture travels, the reference does not:

    python tests/golden/make_golden_nimagenet.py

The reference module n_imagenet/real_cnn_model/data/imagenet.py is IMPORTED (never copied) with the same
in-process stand-ins make_golden.py uses, plus scatter_max / scatter_min for torch_scatter ("parity
unpinned" at that boundary: empty entries 0, as torch_scatter documents for a fresh output).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

REF = mg.REF
NAMES = ["acc", "acc_time", "acc_count", "acc_count_pol", "acc_count_only", "acc_all", "flat", "flat_pol",
         "acc_exp", "acc_time_pol", "acc_intensity"]


ARG_PICK = "first"


def _import_imagenet():
    mg._install_standins()
    ts = sys.modules["torch_scatter"]

    def _scatter_ext(kind):
        def f(src, index, dim=-1, out=None, dim_size=None):
            assert src.dim() == 1 and out is None
            o = torch.zeros(dim_size, dtype=src.dtype).scatter_reduce_(0, index, src, reduce=kind, include_self=False)
            # arg indices as torch_scatter's sequential CPU loop leaves them: the FIRST element attaining the extremum
            # (strict comparison), src.size for slots nothing was scattered into.  ARG_PICK = "last" flips the
            # tie-break; main() checks that no golden depends on it.
            arg = torch.full((dim_size,), src.shape[0], dtype=torch.long)
            order = range(src.shape[0] - 1, -1, -1) if ARG_PICK == "first" else range(src.shape[0])
            sv, iv, ov = src.tolist(), index.tolist(), o.tolist()
            for e in order:                       # later assignments win: walking backwards leaves the first
                if sv[e] == ov[iv[e]]:
                    arg[iv[e]] = e
            return o, arg
        return f

    ts.scatter_max, ts.scatter_min = _scatter_ext("amax"), _scatter_ext("amin")
    os.chdir(REF)
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "representations"))
    spec = importlib.util.spec_from_file_location(
        "ref_imagenet", os.path.join(REF, "n_imagenet", "real_cnn_model", "data", "imagenet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_case(N, sensor_w, sensor_h, W, H, seed, single=None):
    """(N,4) float64 rows [x, y, t_seconds, p] as parse_event hands them over (imagenet.py:45-55,104-108):
    sensor coordinates scaled to the image size (fractional), microsecond timestamps / 1e6, p in {-1,+1}."""
    ev = mg.make_events(N, sensor_w, sensor_h, seed=seed, single_polarity=single).astype(np.float64)
    ev[:, 2] /= 1000000
    ev[:, 0] *= W / sensor_w
    ev[:, 1] *= H / sensor_h
    return ev


def main():
    ref = _import_imagenet()
    g = {}
    cases = [("a", make_case(8000, 640, 480, 128, 96, 701), 96, 128),
             ("b", make_case(3000, 64, 48, 32, 24, 702), 24, 32),            # dense: ~4 events per pixel
             ("pos", make_case(2000, 80, 60, 80, 60, 703, single=1), 60, 80),  # no negative events: 0/0 -> nan
             ("c224", make_case(10000, 640, 480, 224, 224, 704), 224, 224)]
    for tag, ev, H, W in cases:
        g[tag + "_events"], g[tag + "_H"], g[tag + "_W"] = ev, H, W
        names = NAMES if tag != "c224" else ["acc_all", "acc_exp"]
        for name in names:
            out = getattr(ref, "reshape_then_" + name)(torch.from_numpy(ev.copy()), height=H, width=W)
            a = out.numpy()
            assert a.dtype == np.float32, (name, a.dtype)
            g["%s_%s" % (tag, name)] = np.ascontiguousarray(a)
    # DiST (reshape_then_acc_adj_sort, :873-999) on the same cases
    for tag, ev, H, W in cases[:3]:
        g[tag + "_acc_adj_sort"] = ref.reshape_then_acc_adj_sort(torch.from_numpy(ev.copy()), height=H, width=W).numpy()
    # sorted timestamp image (reshape_then_acc_sort, :513-838), strict=False, a few keyword combinations
    base = dict(strict=False, denoise_image=False, denoise_sort=False)
    combos = {"s0": dict(global_time=True, neglect_polarity=True, use_image=True, quantize_sort=None),
              "s1": dict(global_time=True, neglect_polarity=False, use_image=True, quantize_sort=8),
              "s2": dict(global_time=False, neglect_polarity=False, use_image=False, quantize_sort=[4, 16]),
              "s3": dict(global_time=False, neglect_polarity=True, use_image=False, quantize_sort=None)}
    for tag, ev, H, W in cases[:2]:
        for ck, kw in combos.items():
            g["%s_acc_sort_%s" % (tag, ck)] = ref.reshape_then_acc_sort(
                torch.from_numpy(ev.copy()), height=H, width=W, **base, **kw).numpy()
    # strict=True (:563-590,685-748): the dense rank of the per-pixel latest indices; the goldens must not depend
    # on which of several tied events scatter_max's arg names
    global ARG_PICK
    strict_combos = {"t0": dict(global_time=True, neglect_polarity=True, use_image=True, quantize_sort=None),
                     "t1": dict(global_time=True, neglect_polarity=False, use_image=True, quantize_sort=8),
                     "t2": dict(global_time=False, neglect_polarity=False, use_image=False, quantize_sort=[4, 16]),
                     "t3": dict(global_time=False, neglect_polarity=True, use_image=False, quantize_sort=None)}
    sbase = dict(strict=True, denoise_image=False, denoise_sort=False)
    for tag, ev, H, W in cases[:3]:
        for ck, kw in strict_combos.items():
            outs = []
            for pick in ("first", "last"):
                ARG_PICK = pick
                outs.append(ref.reshape_then_acc_sort(torch.from_numpy(ev.copy()), height=H, width=W, **sbase, **kw).numpy())
            assert np.array_equal(outs[0], outs[1]), (tag, ck)
            g["%s_acc_sort_%s" % (tag, ck)] = outs[0]
    ARG_PICK = "first"
    # the empty-tensor substitutions (imagenet.py:258-261,483-486)
    for name in ("acc_count", "acc_time_pol"):
        g["empty_" + name] = getattr(ref, "reshape_then_" + name)(torch.zeros((0, 4), dtype=torch.float64),
                                                                height=24, width=32).numpy()
    # the wrappers of the path's own builders that run in the reference (imagenet.py:1025-1107): integral fields
    for tag, N, W, H, seed in (("w1", 3000, 64, 48, 711), ("w2", 9000, 128, 96, 712)):
        ev = mg.make_events(N, W, H, seed=seed).astype(np.float64)
        g[tag + "_events"], g[tag + "_H"], g[tag + "_W"] = ev, H, W
        for name in ("optimized", "event_stack", "tore"):
            g["%s_%s" % (tag, name)] = getattr(ref, "reshape_then_" + name)(torch.from_numpy(ev.copy()),
                                                                          height=H, width=W).numpy()
    np.savez_compressed(os.path.join(HERE, "nimagenet_acc.npz"), **g)
    print("wrote nimagenet_acc.npz", len(g), "arrays")


if __name__ == "__main__":
    main()
