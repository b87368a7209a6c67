import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from event_representation_study_amd.engine import EventBatch
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
masks = {"all": range(256), "drop cu0": range(1, 256), "drop cu0-1": range(2, 256), "drop cu0-3": range(4, 256), "drop cu0-7": range(8, 256),
         "drop cu248-255": range(0, 248), "drop cu100": [c for c in range(256) if c != 100],
         "drop 1 per 32": [c for c in range(256) if c % 32 != 0], "drop cu0,128": [c for c in range(256) if c not in (0, 128)]}
for name, bits in masks.items():
    s = masked_stream(bits)
    print("%-16s slow %.1f  fast %.1f" % (name, t_on(s, lambda: eb.optimized(out=slow)), t_on(s, lambda: eb.optimized(out=fast))))
