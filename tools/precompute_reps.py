#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: a stream of 1 Mpx windows -> (640, 640, 12) float32 files;
prints the end-to-end output GB/s (events already in host memory, files on --out, default /dev/shm)."""
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
    args = ap.parse_args()
    H, W = 720, 1280
    pool = [make_events(args.events, W, H, seed=9000 + i) for i in range(args.batch)]   # reused: generation is not the subject
    batches = ([pool[i % args.batch] for i in range(args.batch)] for _ in range(args.samples // args.batch))
    pc = RepPrecomputer(H, W, 640, args.builder)
    pc.run([pool], args.out, keep_files=False)                                           # warm-up
    n, nbytes, el = pc.run(batches, args.out, keep_files=args.keep)
    if not args.keep:
        shutil.rmtree(args.out, ignore_errors=True)
    print(json.dumps({"config": "precompute 1280x720 -> (640,640,12) f32 .npy", "builder": args.builder, "samples": n,
                      "events_per_sample": args.events, "seconds": el, "samples_per_s": n / el,
                      "output_GBps": nbytes / el / 1e9, "out": args.out}))


if __name__ == "__main__":
    main()
