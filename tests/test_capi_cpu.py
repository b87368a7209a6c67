"""CPU-only: the C-ABI library loads, exports every symbol include/evrep.h declares, and its
host-side entry points (plan / workspace sizing, argument checks) behave.  No kernel is launched."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from event_representation_study_amd import build, _lib
    build.build()
    return _lib.load()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "evrep.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(evrep_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from event_representation_study_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 14
    assert set(declared) == set(_lib.SYMBOLS), (set(declared) ^ set(_lib.SYMBOLS))
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.evrep_abi_version() == _lib.ABI_VERSION


def test_plan_struct_layout_matches_header(lib):
    from event_representation_study_amd._lib import Plan
    p = Plan()
    assert lib.evrep_plan_init(ctypes.byref(p), 32, 480, 640, 32 * 50000, 50000) == 0
    assert (p.B, p.H, p.W, p.total_events, p.max_events_per_window) == (32, 480, 640, 1600000, 50000)
    assert p.nchunk == 5 and p.chunk >= 256 and p.nblk * p.chunk >= 50000
    offs = [p.off_meta, p.off_table, p.off_stats, p.off_rowoff, p.off_chunkoff, p.off_sorted1, p.off_sorted2,
            p.off_cuts, p.off_scratch]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert lib.evrep_workspace_bytes(ctypes.byref(p)) == p.workspace_bytes > p.off_scratch
    # 2 x 16 B per event for the two partition levels dominate the workspace
    assert p.workspace_bytes >= 2 * 16 * 1600000


@pytest.mark.parametrize("args", [(0, 480, 640, 10, 10), (1, 0, 640, 10, 10), (1, 480, 5000, 10, 10),
                                  (1, 480, 640, 10, 11), (1, 480, 640, -1, 0), (1, 480, 640, 1 << 31, 1),
                                  (65536, 8, 8, 10, 10),            # windows ride gridDim.z
                                  (60000, 4096, 4096, 10, 10)])     # > 2^31 work units
def test_plan_init_rejects_bad_arguments(lib, args):
    from event_representation_study_amd._lib import Plan, EVREP_EINVAL
    assert lib.evrep_plan_init(ctypes.byref(Plan()), *args) == EVREP_EINVAL


def test_large_window_partition_geometry(lib):
    from event_representation_study_amd._lib import Plan
    p = Plan()
    assert lib.evrep_plan_init(ctypes.byref(p), 1, 720, 1280, 1000000, 1000000) == 0
    assert p.nblk <= 128 and p.nblk * p.chunk >= 1000000 and p.chunk % 256 == 0 and p.nchunk == 10


def test_null_and_misaligned_pointers_are_refused_before_any_launch(lib):
    from event_representation_study_amd._lib import Plan, EVREP_EINVAL
    p = Plan()
    assert lib.evrep_plan_init(ctypes.byref(p), 1, 8, 8, 4, 4) == 0
    assert lib.evrep_bin_events(ctypes.byref(p), None, None, None, None) == EVREP_EINVAL
    assert lib.evrep_bin_events(ctypes.byref(p), ctypes.c_void_p(8), ctypes.c_void_p(256), ctypes.c_void_p(256),
                                None) == EVREP_EINVAL   # events not 16-byte aligned
    assert lib.evrep_gwd_padded_l1(None, 1, 4, None, 1, 4, 0.7, None, None, None) == EVREP_EINVAL
    i32 = (ctypes.c_int32 * 2)(0, 0)
    assert lib.evrep_polstats(ctypes.byref(p), ctypes.c_void_p(256), ctypes.c_void_p(256), ctypes.c_void_p(256),
                              ctypes.c_void_p(256), 17, i32, i32, 0.3, ctypes.c_void_p(256), None) == EVREP_EINVAL
    bad = (ctypes.c_int32 * 2)(0, 9)   # unknown statistic
    assert lib.evrep_polstats(ctypes.byref(p), ctypes.c_void_p(256), ctypes.c_void_p(256), ctypes.c_void_p(256),
                              ctypes.c_void_p(256), 2, i32, bad, 0.3, ctypes.c_void_p(256), None) == EVREP_EINVAL
    assert lib.evrep_gwd_scratch_bytes(12500, 14400) > 14 * 14400 * 4


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly (never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from event_representation_study_amd import _lib
    from event_representation_study_amd.engine import EventBatch
    from event_representation_study_amd.representations.optimized_representation import get_optimized_representation
    from event_representation_study_amd.synthetic import make_events, to_structured
    with pytest.raises(_lib.EvrepError):
        EventBatch.from_numpy(np.zeros((4, 4), np.int32), 8, 8)
    with pytest.raises(_lib.EvrepError):
        get_optimized_representation(to_structured(make_events(10, 8, 8)), 10, 8, 8)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "event_representation_study_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libevrep_oracle" not in src and "evrep_oracle.c" not in src, f
