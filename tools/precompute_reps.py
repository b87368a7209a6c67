#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: a stream of 1 Mpx windows -> (640, 640, 12) float32 HDF5 files ("repr"
dataset, as precompute_reps.py:432-435 writes them); prints the end-to-end output GB/s (events already in host
memory, files on --out, default /dev/shm) and the GPU-side rate alone (builder + resize, no D2H, no files)."""
import argparse
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from event_representation_study_amd.precompute import RepPrecomputer  # noqa: E402
from event_representation_study_amd.synthetic import make_events  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--events", type=int, default=200000)
    ap.add_argument("--builder", default="optimized")
    ap.add_argument("--out", default="/dev/shm/evrep_precompute")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--container", default="h5", choices=["h5", "npy"])
    ap.add_argument("--writers", type=int, default=16)
    ap.add_argument("--loaders", type=int, default=3)
    args = ap.parse_args()
    H, W = 720, 1280
    pool = [make_events(args.events, W, H, seed=9000 + i) for i in range(args.batch)]   # reused: generation is not the subject
    batches = ([pool[i % args.batch] for i in range(args.batch)] for _ in range(args.samples // args.batch))
    pc = RepPrecomputer(H, W, 640, args.builder, container=args.container, writers=args.writers, loaders=args.loaders)
    pc.run([pool], args.out, keep_files=False)                                           # warm-up
    n, nbytes, el = pc.run(batches, args.out, keep_files=args.keep)
    if not args.keep:
        shutil.rmtree(args.out, ignore_errors=True)
    import time
    import torch
    for _ in range(3):
        pc.represent(pool)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        pc.represent(pool)
    torch.cuda.synchronize()
    gpu_el = (time.perf_counter() - t0) / (reps * len(pool))
    print(json.dumps({"config": "precompute 1280x720 -> (640,640,12) f32 %s" % args.container, "builder": args.builder,
                      "samples": n, "events_per_sample": args.events, "seconds": el, "samples_per_s": n / el,
                      "output_GBps": nbytes / el / 1e9, "out": args.out,
                      "gpu_only_samples_per_s": 1.0 / gpu_el, "gpu_only_us_per_sample": gpu_el * 1e6,
                      "note": "end to end = H2D of the events + bin + build + resize + D2H (19.7 MB per sample over PCIe) "
                              "+ file writes by 4 host threads; gpu_only = bin + build + resize with the events' H2D"}))


if __name__ == "__main__":
    main()
