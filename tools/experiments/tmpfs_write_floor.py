"""Config 5's host floor, without any GPU work: T threads write one 19.66 MB file per sample (640x640x12 float32, the bytes of
`precompute_reps.py`'s "repr" dataset) into /dev/shm from a resident buffer -- (a) a NEW file per sample (what the pipeline
does: tmpfs allocates and fills fresh pages), (b) the same file rewritten in place (pages exist: the copy alone).
Prints samples/s and GB/s per thread count; the cgroup's CPU quota is what bounds both."""
import os
import sys
import threading
import time

import numpy as np

BYTES = 640 * 640 * 12 * 4
OUT = "/dev/shm/evrep_floor"
os.makedirs(OUT, exist_ok=True)
buf = np.random.default_rng(0).integers(0, 255, BYTES, dtype=np.uint8).tobytes()


def run(threads, per_thread, fresh):
    def work(tid):
        path = os.path.join(OUT, "t%d" % tid)
        for i in range(per_thread):
            p = "%s_%d" % (path, i) if fresh else path
            fd = os.open(p, os.O_WRONLY | os.O_CREAT | (os.O_TRUNC if fresh else 0), 0o600)
            os.write(fd, buf)
            os.close(fd)
            if fresh:
                os.remove(p)
        if not fresh:
            os.remove(path)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    n = threads * per_thread
    return n / dt, n * BYTES / dt / 1e9


quota = "?"
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()
    quota = "%.1f cores" % (int(q) / int(p)) if q != "max" else "none"
except Exception:
    pass
print("cgroup cpu quota:", quota, " host cores:", os.cpu_count())
for fresh in (True, False):
    for threads in (4, 8, 16, 32):
        sps, gbps = run(threads, 24, fresh)
        print("%-34s threads %2d: %7.0f samples/s  %6.1f GB/s" % ("new file per sample (fresh pages)" if fresh else "one file rewritten in place", threads, sps, gbps), flush=True)
