#!/usr/bin/env python3
"""BASELINE.json configs[4]: a stream of 1 Mpx windows -> (640, 640, 12) float32 HDF5 files ("repr" dataset, as
precompute_reps.py:432-435 writes them); prints the end-to-end output GB/s (events already in host memory, files on
--out, default /dev/shm) and the GPU-side rate alone (builder + resize, no D2H, no files).

`--gpus N` (r06): one process per GPU (self-launched under torch.distributed.run, or started by a launcher that sets
RANK / WORLD_SIZE); the sample stream -- or, with --events-h5, the keys of the reference's event container
(precompute_reps.py:408-409) -- is dealt round-robin, every rank writes its samples under their GLOBAL numbers, and rank 0
prints the aggregate (the reference: a Pool of 8 CPU workers, precompute_reps.py:439-466)."""
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
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--events-h5", default=None, help="the reference's input container: flat (n, 4) int32 datasets, one per key")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    args = ap.parse_args()
    import torch
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        raise SystemExit(subprocess.call(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
             "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        return main_ranks(args, rank, world)
    H, W = args.height, args.width
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


def main_ranks(args, rank, world):
    """One rank of N: its share of the stream on its own GPU, figures folded by one all_gather."""
    import torch
    from event_representation_study_amd import h5lite
    from event_representation_study_amd.precompute import aggregate_over_ranks, shard_keys
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    torch.distributed.init_process_group("nccl", device_id=device)
    H, W = args.height, args.width
    pc = RepPrecomputer(H, W, 640, args.builder, device=str(device), container=args.container,
                        writers=max(2, args.writers // world), loaders=args.loaders)
    out = args.out                                   # ONE directory: the global numbering keeps the ranks apart
    os.makedirs(out, exist_ok=True)
    if args.events_h5:
        keys = sorted(h5lite.File(args.events_h5).keys())
        torch.distributed.barrier()
        trip = pc.run_h5(args.events_h5, keys, out, batch=args.batch, rank=rank, world=world, keep_files=args.keep)
    else:
        pool = [make_events(args.events, W, H, seed=9000 + 100 * rank + i) for i in range(args.batch)]
        mine, first, stride = shard_keys(list(range(args.samples)), rank, world)
        pc.run([pool], out + "_warm%d" % rank, keep_files=False)
        shutil.rmtree(out + "_warm%d" % rank, ignore_errors=True)
        torch.distributed.barrier()
        stream = ([pool[i % args.batch] for i in range(len(mine[k:k + args.batch]))] for k in range(0, len(mine), args.batch))
        trip = pc.run(stream, out, keep_files=args.keep, first_index=first, index_stride=stride)
    res = aggregate_over_ranks(*trip, device=device)
    torch.distributed.barrier()
    if rank == 0:
        if not args.keep:
            shutil.rmtree(out, ignore_errors=True)
        res.update(config="precompute %dx%d -> (640,640,C) f32 %s, %d ranks" % (W, H, args.container, world), builder=args.builder)
        print(json.dumps(res))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
