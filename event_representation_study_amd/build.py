"""Build libevrep.so (HIP, gfx950 only) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libevrep.so")
# four translation units, compiled in parallel and linked into one .so (r05; a unity build until then: four minutes)
SOURCES = ["evrep_capi.hip", "evrep_capi_mdes.hip", "evrep_capi_builders.hip", "evrep_capi_gwd.hip"]
_BIN = ["evrep_bin.hip", "evrep_common.h", "evrep_capi_shared.h"]
_BLD = _BIN + ["evrep_builders.hip", "evrep_capi_builders.h"]
UNIT_DEPS = {"evrep_capi.hip": _BIN, "evrep_capi_mdes.hip": _BLD, "evrep_capi_builders.hip": _BLD,
             "evrep_capi_gwd.hip": ["evrep_gwd.hip", "evrep_otmi.hip", "evrep_gw.hip", "evrep_common.h", "evrep_capi_shared.h"]}
OBJDIR = os.path.join(PKG, "_obj")

# -ffp-contract=off: the parity contract is bit-exactness with the reference's separate
# multiply / add / subtract (variance = mean(x^2) - mean(x)^2, src**2, w*p), so no FMA fusion.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libevrep.so cannot be built")
    return exe


def _mtime(d):
    return os.path.getmtime(d if os.path.isabs(d) else os.path.join(CSRC, d))


def _unit_deps(src):
    return [src] + UNIT_DEPS[src] + [os.path.join(ROOT, "include", "evrep.h")]


def _obj(src, tag=""):
    return os.path.join(OBJDIR, os.path.splitext(src)[0] + tag + ".o")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(_mtime(d) > t for src in SOURCES for d in _unit_deps(src))


def build(force=False, verbose=True, extra_flags=(), lib=LIB):
    """Compile the translation units that are out of date (in parallel) and link `lib`.  Objects are cached under _obj/ per
    set of extra flags, so experiment builds (tools/variants) do not disturb the library's own objects."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    if lib == LIB and not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    tag = ("-" + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8]) if extra_flags else ""
    todo = []
    for src in SOURCES:
        o = _obj(src, tag)
        if force or not os.path.exists(o) or any(_mtime(d) > os.path.getmtime(o) for d in _unit_deps(src)):
            todo.append([hipcc()] + FLAGS + list(extra_flags) + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c",
                                                                  os.path.join(CSRC, src), "-o", o])

    def run(cmd):
        if verbose:
            print("[evrep build]", " ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        list(ex.map(run, todo))
    run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [_obj(src, tag) for src in SOURCES])
    return lib


EXAMPLE_SRC = os.path.join(ROOT, "examples", "capi_ergo12.cpp")
EXAMPLE_BIN = os.path.join(ROOT, "examples", "capi_ergo12")


def build_example(force=False, verbose=True):
    """examples/capi_ergo12: the C ABI driven from plain C++ (HIP runtime only, no torch, no Python)."""
    build(verbose=verbose)
    if not force and os.path.exists(EXAMPLE_BIN) and \
            os.path.getmtime(EXAMPLE_BIN) >= max(os.path.getmtime(EXAMPLE_SRC), os.path.getmtime(LIB)):
        return EXAMPLE_BIN
    cmd = [hipcc(), "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), EXAMPLE_SRC, "-L", PKG, "-levrep",
           "-Wl,-rpath,$ORIGIN/../event_representation_study_amd", "-o", EXAMPLE_BIN]
    if verbose:
        print("[evrep build]", " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return EXAMPLE_BIN


if __name__ == "__main__":
    # python -m event_representation_study_amd.build [--force] [-o tools/variants/libevrep_timing.so] [-DEVREP_TIMING ...]
    args = [a for a in sys.argv[1:] if a != "--force"]
    out = LIB
    if "-o" in args:
        i = args.index("-o")
        out = os.path.abspath(args[i + 1])
        os.makedirs(os.path.dirname(out), exist_ok=True)
        del args[i:i + 2]
    build(force="--force" in sys.argv, extra_flags=args, lib=out)
    if out == LIB:
        build_example(force="--force" in sys.argv)
