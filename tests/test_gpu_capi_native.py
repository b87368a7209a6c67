"""-m gpu: the C ABI driven from plain C++ (examples/capi_ergo12.cpp: HIP runtime only, no torch, no
Python) gives the same bytes as the Python engine and as the CPU oracle on the same events."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M64 = (1 << 64) - 1


def splitmix_window(seed, n, H, W):
    """The event generator of examples/capi_ergo12.cpp, restated."""
    s = seed
    ev = np.empty((n, 4), dtype=np.int32)
    t = 0
    for i in range(n):
        s = (s + 0x9E3779B97F4A7C15) & M64
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        r = z ^ (z >> 31)
        t += (r >> 60) & 3
        ev[i] = ((r & 0xffffffff) % W, ((r >> 32) & 0xfffffff) % H, t, 1 if (r >> 62) & 1 else -1)
    return ev


def fnv1a64(buf):
    h = 0xcbf29ce484222325
    # FNV-1a is sequential; fold 8 KiB at a time through Python ints via numpy is not possible -> plain loop on bytes
    for b in memoryview(buf).cast("B").tobytes():
        h = ((h ^ b) * 0x100000001b3) & M64
    return h


def test_native_cpp_caller_matches_engine_and_oracle(oracle):
    from event_representation_study_amd import build
    from event_representation_study_amd.engine import EventBatch
    exe = build.build_example(verbose=False)
    B, N, H, W = 3, 4000, 48, 80
    res = subprocess.run([exe, str(B), str(N), str(H), str(W)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = json.loads(res.stdout.strip().splitlines()[-1])
    assert (got["B"], got["N"], got["H"], got["W"], got["status_or"]) == (B, N, H, W, 0)
    wins = [splitmix_window(b + 1, N, H, W) for b in range(B)]
    ref = np.stack([oracle.ergo12(w, H, W) for w in wins])
    eng = EventBatch.from_numpy(wins, H, W).optimized().cpu().numpy()
    assert np.array_equal(eng.view(np.uint64), ref.view(np.uint64))
    assert got["fnv1a64"] == "%016x" % fnv1a64(np.ascontiguousarray(ref))
