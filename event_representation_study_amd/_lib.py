"""ctypes binding of libevrep.so (the C ABI declared in include/evrep.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load, importing
a builder raises.  (The library is built in-tree by ``event_representation_study_amd.build``.)
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVREP_LIB_PATH") or os.path.join(_PKG, "libevrep.so")  # env override: A/B builds

EVREP_OK, EVREP_EINVAL, EVREP_EWORKSPACE, EVREP_EHIP, EVREP_ENOTBINNED = 0, 1, 2, 3, 4
ST_EMPTY, ST_OOB, ST_UNSORTED, ST_FLAT_TIME, ST_HOT_OVERFLOW = 1, 2, 4, 8, 16
F64, F32 = 0, 1
MAX_CHANNELS = 16
MAX_DIM = 4096
ABI_VERSION = 3

FUNCS = ["timestamp", "polarity", "count", "timestamp_pos", "timestamp_neg", "count_pos", "count_neg"]
AGGS = ["sum", "mean", "max", "variance"]


class Plan(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("total_events", ctypes.c_int64),
        ("max_events_per_window", ctypes.c_int64),
        ("chunk", ctypes.c_int32), ("nblk", ctypes.c_int32),
        ("nchunk", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("flags", ctypes.c_int32), ("pacing", ctypes.c_int32),
        ("off_meta", ctypes.c_size_t), ("off_table", ctypes.c_size_t), ("off_stats", ctypes.c_size_t),
        ("off_rowoff", ctypes.c_size_t),
        ("off_chunkoff", ctypes.c_size_t),
        ("off_sorted1", ctypes.c_size_t), ("off_sorted2", ctypes.c_size_t), ("off_cuts", ctypes.c_size_t),
        ("off_scratch", ctypes.c_size_t),
        ("workspace_bytes", ctypes.c_size_t),
    ]


# every symbol include/evrep.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _f64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_float
_PP = ctypes.POINTER(Plan)
_I32P = ctypes.POINTER(ctypes.c_int32)
SYMBOLS = {
    "evrep_abi_version": (ctypes.c_int, []),
    "evrep_last_hip_error": (ctypes.c_char_p, []),
    "evrep_plan_init": (ctypes.c_int, [_PP, _i32, _i32, _i32, _i64, _i64]),
    "evrep_plan_init_ex": (ctypes.c_int, [_PP, _i32, _i32, _i32, _i64, _i64, ctypes.c_uint32]),
    "evrep_plan_set_pacing": (ctypes.c_int, [_PP, _i32]),
    "evrep_workspace_bytes": (ctypes.c_size_t, [_PP]),
    "evrep_bin_events": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp]),
    "evrep_probe_store": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp]),
    "evrep_mdes": (ctypes.c_int, [_PP, _vp, _vp, _vp, _i32, _I32P, _I32P, _I32P, _f64, _i32, _vp, _vp]),
    "evrep_mdes_sbt_windows": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "evrep_mdes_ex": (ctypes.c_int, [_PP, _vp, _vp, _vp, _i32, _I32P, _I32P, _I32P, _f64, _i32, _vp, _vp, _vp, _vp]),
    "evrep_optimized": (ctypes.c_int, [_PP, _vp, _vp, _vp, _f64, _i32, _vp, _vp]),
    "evrep_event_stack": (ctypes.c_int, [_PP, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp]),
    "evrep_time_surface": (ctypes.c_int, [_PP, _vp, _vp, _vp, _i32, _vp, _f64, _i32, _f64, _i32, _vp, _vp]),
    "evrep_time_surface_ftime": (ctypes.c_int, [_PP, _vp, _vp, _vp, _i32, _vp, _vp, _f64, _i32, _f64, _i32, _vp, _vp]),
    "evrep_tore": (ctypes.c_int, [_PP, _vp, _vp, _vp, _i32, _i32, _vp, _f32, _vp, _vp]),
    "evrep_tore_ftime": (ctypes.c_int, [_PP, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _f32, _vp, _vp]),
    "evrep_voxel": (ctypes.c_int, [_PP, _vp, _vp, _vp, _i32, _i32, _f64, _vp, _vp]),
    "evrep_voxel_range": (ctypes.c_int, [_PP, _vp, _vp, _vp, _i32, _i32, _f64, _vp, _vp, _vp]),
    "evrep_voxel_tnorm": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp, _i32, _f64, _vp, _vp]),
    "evrep_voxel_subpixel": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "evrep_polstats": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _f64, _vp, _vp]),
    "evrep_est_voxel": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _f64, _f64, _vp, _vp]),
    "evrep_read_status": (ctypes.c_int, [_PP, _vp, _vp, _vp]),
    "evrep_read_bbox": (ctypes.c_int, [_PP, _vp, _vp, _vp]),
    "evrep_copy_window_meta_async": (ctypes.c_int, [_PP, _vp, _vp, _vp]),
    "evrep_gwd_scratch_bytes": (ctypes.c_size_t, [_i64, _i64]),
    "evrep_resize_taps": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _i32, _vp, _vp]),
    "evrep_gw_scratch_bytes": (ctypes.c_size_t, [_i64, _i64, _i32]),
    "evrep_entropic_gw": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _f64, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "evrep_gwd_padded_l1": (ctypes.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _f64, _vp, _vp, _vp]),
    "evrep_gwd_batch_scratch_bytes": (ctypes.c_size_t, [_i32, _i32, _i32, _i64, _i64]),
    "evrep_gwd_padded_l1_batch": (ctypes.c_int, [_i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _i64, _f64, _vp, _vp, _vp]),
    "evrep_otmi_scratch_bytes": (ctypes.c_size_t, [_i32]),
    "evrep_otmi_event_clouds": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "evrep_otmi_rep_clouds": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
}

# evrep_plan_init_ex flags.  The C library reads no environment variable; the A/B switches of the tests and tools
# are translated here, when a plan is made.
PLAN_NO_KEY_PASS, PLAN_THREE_KERNEL, PLAN_FORCE_KEY_SORTED, PLAN_BIG_BLOCKS, PLAN_NO_FUSED_SCATTER = 1, 2, 4, 8, 16
_ENV_FLAGS = (("EVREP_BIN_CLASSIC", PLAN_NO_KEY_PASS), ("EVREP_BIN_THREE_KERNEL", PLAN_THREE_KERNEL),
              ("EVREP_BIN_KEY_SORTED", PLAN_FORCE_KEY_SORTED), ("EVREP_KS_BIG_BLOCKS", PLAN_BIG_BLOCKS),
              ("EVREP_NO_FUSED_SCATTER", PLAN_NO_FUSED_SCATTER), ("EVREP_X_SPAN2", 64), ("EVREP_X_STAGE128", 128), ("EVREP_X_TAIL_MERGE", 256), ("EVREP_X_STAGE64", 512), ("EVREP_X_HANDOVER2", 1024), ("EVREP_X_HANDOVER_DENSE", 2048), ("EVREP_X_NO_SWEEP_MAIN", 4096), ("EVREP_X_NO_MONSTER_HANDOVER", 8192), ("EVREP_X_VOXEL_ORDERED", 16384), ("EVREP_X_TORE_ORDERED", 32768), ("EVREP_X_POLSTATS_ORDERED", 65536), ("EVREP_X_ESTACK_ORDERED", 131072), ("EVREP_X_MDES_ORDERED", 262144), ("EVREP_X_MDES_STREAM", 524288), ("EVREP_X_TS_STREAM", 1048576), ("EVREP_X_TS_ORDERED", 2097152), ("EVREP_X_MDES_NO_COOP", 4194304))


# (the per-sample wrappers translate the switches on every call -- a test may flip one between two samples: read CPython's own
#  dictionary behind os.environ instead of twenty-odd os.environ.get calls, 10 us of a 100 us sample)
_ENV_DATA = getattr(os.environ, "_data", None)
try:
    _ENV_KEYS = tuple((os.environ.encodekey(name), bit) for name, bit in _ENV_FLAGS) if isinstance(_ENV_DATA, dict) else None
    _PACING_KEY = os.environ.encodekey("EVREP_PACING") if _ENV_KEYS is not None else None
except AttributeError:
    _ENV_KEYS = _PACING_KEY = None


def plan_flags_from_env():
    flags = 0
    if _ENV_KEYS is not None:
        data = _ENV_DATA
        for key, bit in _ENV_KEYS:
            if data.get(key):
                flags |= bit
        return flags
    for name, bit in _ENV_FLAGS:
        if os.environ.get(name):
            flags |= bit
    return flags


def pacing_from_env():
    """EVREP_PACING: -1 automatic (default), 0 off, > 0 hold in 10 ns ticks (A/B timing only)."""
    if _PACING_KEY is not None and _PACING_KEY not in _ENV_DATA:
        return None
    v = os.environ.get("EVREP_PACING")
    return int(v) if v not in (None, "") else None


_lib = None


class EvrepError(RuntimeError):
    pass


def load():
    """Load libevrep.so; raises (loudly) if the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; it must be the process's HIP runtime (same SONAME as
    # /opt/rocm's), otherwise device pointers / streams from torch are foreign to our launches.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise EvrepError(
            "libevrep.so is missing (%s). Build it with `python -m event_representation_study_amd.build`; "
            "there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.evrep_abi_version.restype = ctypes.c_int
    lib.evrep_abi_version.argtypes = []
    if lib.evrep_abi_version() != ABI_VERSION and not (os.environ.get("EVREP_LIB_PATH") and os.environ.get("EVREP_ABI_ANY")):   # (before the symbols are bound: a stale library says so, not AttributeError; EVREP_ABI_ANY: A/B timing of an older build through EVREP_LIB_PATH, tools only)
        raise EvrepError("libevrep.so ABI %d != binding ABI %d: rebuild with `python -m event_representation_study_amd.build`"
                         % (lib.evrep_abi_version(), ABI_VERSION))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc == EVREP_OK:
        return
    names = {EVREP_EINVAL: "EVREP_EINVAL (bad argument)", EVREP_EWORKSPACE: "EVREP_EWORKSPACE",
             EVREP_EHIP: "EVREP_EHIP", EVREP_ENOTBINNED: "EVREP_ENOTBINNED"}
    msg = names.get(rc, "rc=%d" % rc)
    if rc == EVREP_EHIP:
        msg += ": " + (load().evrep_last_hip_error() or b"").decode()
    raise EvrepError("%s failed: %s" % (what or "libevrep call", msg))


def int32_array(values):
    return (ctypes.c_int32 * len(values))(*[int(v) for v in values])
