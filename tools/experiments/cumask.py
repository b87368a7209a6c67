"""CU-masked streams: the binning pass of batch k+1 on a few reserved CUs while the builder of batch k runs on the rest."""
import ctypes, os, sys, time
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
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
H, W, N, B = 480, 640, 50000, 32
ebs = [EventBatch.from_numpy([make_events(N, W, H, seed=100 * j + i) for i in range(B)], H, W) for j in range(2)]
out, us, _ = probe_output_placement((B, H, W, 12), torch.float64, candidates=16)
outs = [out, probe_output_placement((B, H, W, 12), torch.float64, candidates=16)[0]]
print("placement", us)
def t_on(stream, fn, n=200):
    with torch.cuda.stream(stream):
        for _ in range(20): fn()
        stream.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(n): fn()
        b.record(stream); stream.synchronize()
    return a.elapsed_time(b) / n * 1e3
allc = list(range(256))
pattern = sys.argv[1] if len(sys.argv) > 1 else "low"
for nbin in (0, 24, 32, 40, 48):
    if pattern == "low": binc = allc[:nbin]
    elif pattern == "stride": binc = [c for c in allc if (c % (256 // nbin) == 0)] if nbin else []
    else: binc = [c for c in allc if (c % 32) < nbin // 8] if nbin else []          # the first nbin/8 CUs of every group of 32
    buildc = [c for c in allc if c not in binc]
    sb = masked_stream(buildc)
    tb = t_on(sb, lambda: ebs[0].optimized(out=outs[0]))
    if nbin:
        sk = masked_stream(binc)
        tk = t_on(sk, lambda: ebs[1].rebin())
    else:
        sk, tk = sb, t_on(sb, lambda: ebs[0].rebin())
    # pipelined: bin(k+1) on sk || build(k) on sb
    steps = 400
    torch.cuda.synchronize()
    def run(steps):
        evb = [None, None]; evd = [None, None]
        for k in range(steps):
            j = k & 1
            with torch.cuda.stream(sk):
                if evd[j] is not None: sk.wait_event(evd[j])
                ebs[j].rebin()
                e = torch.cuda.Event(); e.record(sk); evb[j] = e
            if k > 0:
                jp = (k - 1) & 1
                with torch.cuda.stream(sb):
                    sb.wait_event(evb[jp])
                    ebs[jp].optimized(out=outs[jp])
                    e = torch.cuda.Event(); e.record(sb); evd[jp] = e
        torch.cuda.synchronize()
    run(40)
    t0 = time.perf_counter(); run(steps); dt = (time.perf_counter() - t0) / steps * 1e6
    print("pattern %s bin CUs %3d: builder alone %.1f us, bin alone %.1f us, pipelined step %.1f us" % (pattern, nbin, tb, tk, dt))
