"""Builder launch time vs the number of CUs its stream may use (hipExtStreamCreateWithCUMask), on a slow and a fast placement."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from event_representation_study_amd.engine import EventBatch, probe_output_placement
from event_representation_study_amd.synthetic import make_events
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = [0] * 8
    for b in bits: words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * 8)(*words)
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr) == 0
    return torch.cuda.ExternalStream(s.value)
H, W, N, B = 480, 640, 50000, 32
eb = EventBatch.from_numpy([make_events(N, W, H, seed=i) for i in range(B)], H, W)
cands = [torch.empty((B, H, W, 12), dtype=torch.float64, device="cuda:0") for _ in range(16)]
def t_on(stream, fn, n=100):
    with torch.cuda.stream(stream):
        for _ in range(10): fn()
        stream.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(n): fn()
        b.record(stream); stream.synchronize()
    return a.elapsed_time(b) / n * 1e3
full = masked_stream(range(256))
times = [t_on(full, lambda o=o: eb.optimized(out=o), 30) for o in cands]
fast, slow = cands[min(range(16), key=lambda i: times[i])], cands[max(range(16), key=lambda i: times[i])]
print("placements: fast %.1f slow %.1f" % (min(times), max(times)))
mode = sys.argv[1] if len(sys.argv) > 1 else "high"
for ncu in (256, 248, 240, 232, 224, 208, 192, 160, 128):
    if mode == "high": bits = range(256 - ncu, 256)          # drop the LOWEST-numbered CUs
    elif mode == "low": bits = range(0, ncu)                 # drop the highest-numbered
    else: bits = [c for c in range(256) if (c % 32) < ncu // 8]   # the same number from every group of 32
    s = masked_stream(bits)
    print("mode %s  %3d CUs: slow placement %.1f us, fast placement %.1f us, bin %.1f us" % (
        mode, ncu, t_on(s, lambda: eb.optimized(out=slow)), t_on(s, lambda: eb.optimized(out=fast)), t_on(s, lambda: eb.rebin())))
