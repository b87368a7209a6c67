"""Offline precompute of representations (SURVEY.md 8 row F2 / BASELINE config 5): the GPU-side
counterpart of ``Prophesee.process_representations`` (ev-YOLOv6/yolov6/data/gen4/precompute_reps.py:
405-466): events (n, 4) -> representation -> forced (S, S) per-channel resize -> float32 file per sample.

Differences, stated plainly: (1) h5py is absent in this image, so samples are written as ``.npy``
(the reference keeps that very call as a comment, precompute_reps.py:431) -- same array, same dtype,
different container; (2) the resize is the OpenCV INTER_AREA / INTER_LINEAR restatement of
``gwd_pipeline`` (parity unpinned against cv2); (3) instead of a Pool(8) of CPU workers, windows are
batched through one GPU and a small thread pool only drains pinned host buffers to disk.
"""
import os
import queue
import threading
import time

import numpy as np
import torch

from .engine import EventBatch
from .gwd_pipeline import area_weights, linear_weights


class RepPrecomputer:
    def __init__(self, height, width, out_size=640, builder="optimized", device="cuda:0", writers=4):
        self.H, self.W, self.S = int(height), int(width), int(out_size)
        self.builder = builder
        self.device = torch.device(device)
        r = self.S / max(self.H, self.W)
        fn = area_weights if r < 1 else linear_weights            # resize_image_process: area when shrinking
        self.wy = torch.from_numpy(fn(self.H, self.S)).to(self.device)
        self.wx = torch.from_numpy(fn(self.W, self.S)).to(self.device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.nwriters = writers

    def _build(self, batch):
        if self.builder == "optimized":
            return batch.optimized(scale=255.0)                   # get_item_transform scales by 255
        if self.builder == "event_stack":
            return batch.event_stack(12, premap=True, scale=255.0).to(torch.float64)
        if self.builder == "time_surface":
            return batch.time_surface(6, 50000.0, premap=True, scale=255.0)
        if self.builder == "tore":
            return batch.tore(6, frame_mode=1, scale=255.0).to(torch.float64)
        raise ValueError(self.builder)

    def run(self, window_batches, out_dir, keep_files=True):
        """window_batches: iterable of lists of (n, 4) int32 arrays.  Returns (samples, bytes, seconds)."""
        os.makedirs(out_dir, exist_ok=True)
        q = queue.Queue(maxsize=4)
        written = [0]

        def writer():
            while True:
                item = q.get()
                if item is None:
                    return
                idx0, host, ev = item
                ev.synchronize()                                   # the D2H copy of this buffer has landed
                for k in range(host.shape[0]):
                    path = os.path.join(out_dir, "%d.npy" % (idx0 + k))
                    np.save(path, host[k].numpy())
                    written[0] += host[k].numel() * 4
                    if not keep_files:
                        os.remove(path)

        threads = [threading.Thread(target=writer, daemon=True) for _ in range(self.nwriters)]
        for t in threads:
            t.start()
        t0 = time.perf_counter()
        count = 0
        for wins in window_batches:
            batch = EventBatch.from_numpy(wins, self.H, self.W, device=self.device)
            rep = self._build(batch)                                # (B, H, W, C) float64 on the GPU
            small = torch.einsum("yi,bijc,xj->byxc", self.wy, rep, self.wx).to(torch.float32).contiguous()
            host = torch.empty(small.shape, dtype=torch.float32, pin_memory=True)
            done = torch.cuda.Event()
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.copy_stream):
                host.copy_(small, non_blocking=True)
                small.record_stream(self.copy_stream)
                done.record(self.copy_stream)
            q.put((count, host, done))
            count += len(wins)
        for _ in threads:
            q.put(None)
        for t in threads:
            t.join()
        el = time.perf_counter() - t0
        return count, written[0], el
