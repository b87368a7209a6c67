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


def _flag_hash(extra_flags=()):
    import hashlib
    return hashlib.sha1(" ".join(FLAGS + list(extra_flags)).encode()).hexdigest()[:16]


def _stamp(lib):
    return lib + ".flags"


def needs_build(extra_flags=(), lib=LIB):
    """Out of date if missing, older than a source, or built with OTHER flags (a library left behind by an experiment
    build with -D switches is not silently kept: its flag hash is recorded beside it)."""
    if not os.path.exists(lib):
        return True
    try:
        if open(_stamp(lib)).read().strip() != _flag_hash(extra_flags):
            return True
    except OSError:
        pass        # a library that travelled without its stamp (the GPU box gets the prebuilt .so): trust the mtimes
    t = os.path.getmtime(lib)
    return any(_mtime(d) > t for src in SOURCES for d in _unit_deps(src))


class _BuildLock:
    """One builder at a time per checkout (pytest workers, one process per GPU): an exclusive flock on _obj/.lock."""
    def __enter__(self):
        import fcntl
        os.makedirs(OBJDIR, exist_ok=True)
        self.f = open(os.path.join(OBJDIR, ".lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def build(force=False, verbose=True, extra_flags=(), lib=LIB):
    """Compile the translation units that are out of date (in parallel) and link `lib`.  Objects are cached under _obj/ per
    set of extra flags, so experiment builds (tools/variants) do not disturb the library's own objects."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    extra_flags = list(extra_flags)
    if not force and not needs_build(extra_flags, lib):
        return lib
    with _BuildLock():
        if not force and not needs_build(extra_flags, lib):      # another process built it while this one waited
            return lib
        tag = ("-" + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8]) if extra_flags else ""
        todo = []
        for src in SOURCES:
            o = _obj(src, tag)
            if force or not os.path.exists(o) or any(_mtime(d) > os.path.getmtime(o) for d in _unit_deps(src)):
                todo.append((o, [hipcc()] + FLAGS + extra_flags + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c",
                                                                   os.path.join(CSRC, src)]))

        def run(cmd):
            if verbose:
                print("[evrep build]", " ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)

        def compile_one(job):
            # into a temporary name, renamed when hipcc succeeded: an interrupted compile never leaves a partial object
            # that looks up to date
            o, cmd = job
            tmp = "%s.tmp%d" % (o, os.getpid())
            try:
                run(cmd + ["-o", tmp])
                os.replace(tmp, o)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)

        with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
            list(ex.map(compile_one, todo))
        tmp = "%s.tmp%d" % (lib, os.getpid())
        try:
            run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [_obj(src, tag) for src in SOURCES])
            os.replace(tmp, lib)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        with open(_stamp(lib), "w") as f:
            f.write(_flag_hash(extra_flags) + "\n")
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
