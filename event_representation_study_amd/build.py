"""Build libevrep.so (HIP, gfx950 only) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libevrep.so")
SOURCES = ["evrep_capi.hip"]                      # unity build: includes the kernel files
DEPS = ["evrep_capi.hip", "evrep_bin.hip", "evrep_builders.hip", "evrep_gwd.hip", "evrep_otmi.hip", "evrep_gw.hip", "evrep_common.h",
        os.path.join(ROOT, "include", "evrep.h")]

# -ffp-contract=off: the parity contract is bit-exactness with the reference's separate
# multiply / add / subtract (variance = mean(x^2) - mean(x)^2, src**2, w*p), so no FMA fusion.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libevrep.so cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for d in DEPS:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=True, extra_flags=()):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + list(extra_flags) + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[evrep build]", " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


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
    build(force="--force" in sys.argv, extra_flags=[a for a in sys.argv[1:] if a != "--force"])
    build_example(force="--force" in sys.argv)
