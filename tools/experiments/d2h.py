import torch, time, numpy as np, os, threading
x = torch.empty((8, 640, 640, 12), dtype=torch.float32, device="cuda")
h = [torch.empty(x.shape, dtype=torch.float32, pin_memory=True) for _ in range(4)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(40): h[i % 4].copy_(x, non_blocking=True)
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("D2H pinned GB/s: %.1f" % (40 * x.numel() * 4 / el / 1e9))
a = h[0][0].numpy()
os.makedirs("/dev/shm/t", exist_ok=True)
def wr(k, n):
    for i in range(n):
        with open("/dev/shm/t/%d_%d" % (k, i), "wb") as f: f.write(memoryview(a).cast("B"))
        os.remove("/dev/shm/t/%d_%d" % (k, i))
for nt in (1, 4, 16):
    ths = [threading.Thread(target=wr, args=(k, 16)) for k in range(nt)]
    t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]; el = time.perf_counter() - t0
    print("shm write %d threads GB/s: %.1f" % (nt, nt * 16 * a.nbytes / el / 1e9))
e = torch.empty((1600000, 4), dtype=torch.int32, pin_memory=True); d = torch.empty((1600000, 4), dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(40): d.copy_(e, non_blocking=True)
torch.cuda.synchronize(); print("H2D pinned GB/s: %.1f" % (40 * e.numel() * 4 / (time.perf_counter() - t0) / 1e9))
