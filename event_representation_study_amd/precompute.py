"""Offline precompute of representations (SURVEY.md 8 row F2 / BASELINE config 5): the GPU-side counterpart of
``Prophesee.process_representations`` / ``process_representation``
(ev-YOLOv6/yolov6/data/gen4/precompute_reps.py:405-466):

    events (n, 4) int32 [x, y, t, p]  ->  get_item_transform (x255)  ->  resize_image_process: forced (S, S)
    per-channel cv2.resize (INTER_AREA when shrinking, :216-260)  ->  float32  ->  ``<counter>.h5`` with one
    dataset "repr" (:432-435), which gen4_2yolo.py:383-386 reads back with h5py.

Here windows are batched through one GPU: binning pass + builder (``EventBatch``), ONE launch of the tap-table
resize kernel (``evrep_resize_taps``: every source element read once, float32 written once -- not two dense
float64 GEMMs with (S x H) / (S x W) weight matrices), pinned-buffer D2H on a copy stream, and a small thread
pool that writes real HDF5 files (``h5lite``: header + array bytes).  Inputs can come straight from the reference's
event containers (``run_h5``: flat (n, 4) int32 datasets under string keys, :408-409).

Differences, stated plainly: (1) the resize is the OpenCV INTER_AREA / INTER_LINEAR restatement of
``gwd_pipeline`` -- parity unpinned against cv2, which is absent; (2) instead of a Pool(8) of CPU workers
(``process_representations``, :439-466; ``evlicious.tools.TaskManager``, task_manager.py:8-44) the windows of a rank share
its GPU and the CPU only drains buffers to disk; on N GPUs the sample keys are dealt round-robin to one process per GPU
(``run_h5(rank=, world=)``: sample ``i`` of the key list is written as ``<i>.h5`` by rank ``i mod N`` -- the same
``counter`` numbering as the reference's, no data-path collective), and ``aggregate_over_ranks`` sums samples and bytes and
takes the slowest rank's seconds (one all_gather of three scalars).  As in the reference, the resize is skipped when
the representation's long side already equals S (``if r != 1``, :228), and TORE is built on the events' bounding
box per sample (gen4_transforms' branch), so its samples are resized one by one.
"""
import collections
import concurrent.futures
import os
import queue
import threading
import time

import numpy as np
import torch

from . import h5lite
from .engine import EventBatch
from .gwd_pipeline import resize_batch

BUILDERS = ("optimized", "event_stack", "time_surface", "tore", "voxel_grid")


class RepPrecomputer:
    def __init__(self, height, width, out_size=640, builder="optimized", device="cuda:0", writers=4, container="h5",
                 augment=False, loaders=3):
        if builder not in BUILDERS:
            raise ValueError("builder must be one of %r" % (BUILDERS,))
        if container not in ("h5", "npy"):
            raise ValueError("container must be 'h5' or 'npy'")
        self.H, self.W, self.S = int(height), int(width), int(out_size)
        self.builder = builder
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.nwriters = int(writers)
        self.container = container
        self.augment = bool(augment)
        self._ring, self._ring_lock = {}, threading.Lock()
        self.nloaders = int(loaders)
        self._in_ring, self._in_lock = [], threading.Lock()

    # ------------------------------------------------------------------------------------------ GPU side
    def _build(self, batch):
        """(B, H, W, C) tensor, already x255 as get_item_transform returns it; TORE: list of (Hbb, Wbb, C)."""
        if self.builder == "optimized":
            return batch.optimized(scale=255.0)
        if self.builder == "event_stack":
            return batch.event_stack(12, premap=True, scale=255.0)
        if self.builder == "time_surface":
            return batch.time_surface(6, 50000.0, premap=True, scale=255.0)
        if self.builder == "voxel_grid":
            return batch.voxel(12, mode=1, scale=255.0)
        return batch.tore(6, frame_mode=0, scale=255.0)           # per-sample bounding-box frames

    def _interp(self, h0, w0):
        r = self.S / max(h0, w0)
        return None if r == 1 else ("area" if (r < 1 and not self.augment) else "linear")

    def represent(self, wins):
        """windows -> (B, S, S, C) float32 on the GPU (or (B, H, W, C) when no resize applies)."""
        return self.represent_batch(EventBatch.from_numpy(wins, self.H, self.W, device=self.device))

    def represent_batch(self, batch):
        rep = self._build(batch)
        if isinstance(rep, list):                                  # TORE: its own frame per sample
            outs = []
            for r in rep:
                mode = self._interp(int(r.shape[0]), int(r.shape[1]))
                outs.append(r.to(torch.float32) if mode is None else
                            resize_batch(r[None], self.S, self.S, mode, out_dtype=torch.float32)[0])
            return torch.stack(outs) if len({tuple(o.shape) for o in outs}) == 1 else outs
        mode = self._interp(self.H, self.W)
        if mode is None:
            return rep.to(torch.float32)
        return resize_batch(rep, self.S, self.S, mode, out_dtype=torch.float32)

    # ------------------------------------------------------------------------------------------ host side
    def _pinned_parts(self, shapes):
        """Pinned staging for ONE batch: a flat float32 buffer from a pool (page-locking a fresh 150 MB buffer per batch
        costs more than the copy), cut into one view per part -- so the parts of a batch never share storage, however
        many there are (TORE hands one part per sample when the bounding-box frames differ).  A buffer is handed out again
        only after the writer that drained it has RELEASED it (r04: the ring used to rely on writers finishing in FIFO
        order -- a writer stalled on a slow file could have had its buffer overwritten); when every buffer is still in
        use a new one is page-locked (the writer queue bounds how many can be: 4 queued + one per writer + one filling).
        Returns (views, token); release with ``self._release(token)``."""
        sizes = [int(np.prod(sh)) for sh in shapes]
        total = max(sum(sizes), 1)
        with self._ring_lock:
            pool = self._ring.setdefault("flat", [])
            slot = next((sl for sl in pool if not sl["busy"] and sl["buf"].numel() >= total), None)
            if slot is None:
                slot = next((sl for sl in pool if not sl["busy"]), None)
                if slot is not None:
                    slot["buf"] = torch.empty(total, dtype=torch.float32, pin_memory=True)   # a larger batch than before
            if slot is None:
                slot = {"buf": torch.empty(total, dtype=torch.float32, pin_memory=True), "busy": False}
                pool.append(slot)
            slot["busy"] = True
        flat = slot["buf"]
        views, o = [], 0
        for sh, n in zip(shapes, sizes):
            views.append(flat[o:o + n].view(sh))
            o += n
        return views, slot

    def _release(self, slot):
        with self._ring_lock:
            slot["busy"] = False

    def reserve(self, shapes):
        """Fill the pinned pool up front for batches of these part shapes (page-locking ~150 MB per buffer costs tens of
        milliseconds each: a short run would otherwise spend most of its time in it)."""
        depth = 4 + self.nwriters + 2
        held = []
        while len(self._ring.get("flat", [])) < depth:
            held.append(self._pinned_parts(shapes)[1])
        for slot in held:
            self._release(slot)

    def _write(self, path_base, arr):
        if self.container == "h5":
            h5lite.write_dataset_file(path_base + ".h5", "repr", arr)   # fh.create_dataset("repr", ..., dtype="f4")
            return path_base + ".h5"
        np.save(path_base + ".npy", arr)
        return path_base + ".npy"

    def run(self, window_batches, out_dir, keep_files=True, first_index=0, index_stride=1):
        """window_batches: iterable of lists of (n, 4) int32 arrays.  One file ``<counter>.h5`` per sample; sample k of
        the stream is numbered ``first_index + k * index_stride`` (a rank of N holding every N-th sample: first_index =
        rank, index_stride = N).  Returns (samples, bytes written, seconds)."""
        index_stride = int(index_stride)
        if index_stride < 1:
            raise ValueError("index_stride must be >= 1")
        os.makedirs(out_dir, exist_ok=True)
        q = queue.Queue(maxsize=4)
        written = [0] * self.nwriters                              # one counter per writer thread: no shared update
        errors = []

        def writer(slot):
            while True:
                item = q.get()
                if item is None:
                    return
                idx0, host, ev, token = item
                try:
                    ev.synchronize()                               # the D2H copy of this buffer has landed
                    for k in range(len(host)):
                        arr = host[k].numpy()
                        path = self._write(os.path.join(out_dir, "%d" % (idx0 + k * index_stride)), arr)
                        written[slot] += arr.nbytes
                        if not keep_files:
                            os.remove(path)
                except Exception as e:                             # surfaced by run(): a lost sample is not silent
                    errors.append(e)
                finally:
                    self._release(token)                           # only now may the pinned buffer be handed out again

        threads = [threading.Thread(target=writer, args=(i,), daemon=True) for i in range(self.nwriters)]
        for t in threads:
            t.start()
        t0 = time.perf_counter()
        count = 0
        # loader stage: a few threads turn the next batches (lists of arrays, or callables that read them) into one
        # pinned (total, 4) int32 buffer + offsets while the GPU works on the current one; numpy's copies and the
        # file reads release the GIL.  The main thread only enqueues: H2D, bin, build, resize, D2H.
        main = torch.cuda.current_stream(self.device)
        it = iter(window_batches)
        pending = collections.deque()
        try:
            with concurrent.futures.ThreadPoolExecutor(self.nloaders) as ex:
                def submit_next():
                    try:
                        item = next(it)
                    except StopIteration:
                        return
                    pending.append(ex.submit(self._load, item))
                for _ in range(self.nloaders + 1):
                    submit_next()
                while pending:
                    slot, total, offs, nwin, nmax = pending.popleft().result()
                    submit_next()
                    try:
                        ev_dev = torch.empty((total, 4), dtype=torch.int32, device=self.device)
                        ev_dev.copy_(slot["buf"][:total], non_blocking=True)
                        slot["free"] = torch.cuda.Event()
                        slot["free"].record(main)                          # the loader may refill the buffer after this
                    finally:
                        slot["busy"] = False                               # also when the H2D failed: the slot is not leaked
                    batch = EventBatch(ev_dev, torch.from_numpy(offs), self.H, self.W, max_events_per_window=nmax)
                    small = self.represent_batch(batch)
                    parts = small if isinstance(small, list) else [small]
                    hosts, token = self._pinned_parts([tuple(part.shape) for part in parts])
                    done = torch.cuda.Event()
                    self.copy_stream.wait_stream(main)
                    with torch.cuda.stream(self.copy_stream):
                        for part, host in zip(parts, hosts):
                            host.copy_(part, non_blocking=True)
                            part.record_stream(self.copy_stream)
                        done.record(self.copy_stream)
                    host_list = hosts[0] if not isinstance(small, list) else hosts
                    q.put((first_index + count * index_stride, host_list, done, token))
                    count += nwin
        finally:
            # whatever happened above (a loader raising, a GPU error, out of memory): the writers get their sentinels and
            # are joined, so nothing blocks in q.get() and what was queued is written and reported
            for _ in threads:
                q.put(None)
            for t in threads:
                t.join()
        if errors:
            raise errors[0]
        return count, sum(written), time.perf_counter() - t0

    def _load(self, item):
        """Loader thread: windows -> one pinned (total, 4) int32 buffer (recycled) + int64 offsets."""
        wins = item() if callable(item) else item
        ws = [np.ascontiguousarray(w, dtype=np.int32).reshape(-1, 4) for w in wins]
        offs = np.zeros(len(ws) + 1, dtype=np.int64)
        np.cumsum([w.shape[0] for w in ws], out=offs[1:])
        total = int(offs[-1])
        with self._in_lock:
            slot = None
            for s in self._in_ring:
                if not s["busy"] and s["cap"] >= total:
                    slot = s
                    break
            if slot is None:
                cap = max(total, 1)
                slot = {"buf": torch.empty((cap, 4), dtype=torch.int32, pin_memory=True), "cap": cap, "busy": False, "free": None}
                self._in_ring.append(slot)
            slot["busy"] = True
        if slot["free"] is not None:
            slot["free"].synchronize()                             # its previous H2D has been consumed
        dst = slot["buf"].numpy()
        for w, o in zip(ws, offs[:-1]):
            dst[o:o + w.shape[0]] = w
        return slot, total, offs, len(ws), int(max([w.shape[0] for w in ws], default=0))

    def run_h5(self, event_h5_file, keys, out_dir, batch=8, rank=None, world=None, **kw):
        """The reference's input side: every ``key`` of ``event_h5_file`` is a flat (n, 4) int32 [x, y, t, p] dataset
        (precompute_reps.py:408-409 reads it with np.array(hf.get(key)); fix_events_training views it as '<i4'
        fields, :737-740).  Samples are numbered in key order.

        Data-parallel form (the reference's Pool(8), precompute_reps.py:443): with ``world`` > 1 ranks (default: the
        initialised torch.distributed group) this rank takes every world-th key starting at ``rank`` and writes them
        under their GLOBAL numbers, so N ranks together leave exactly the files one rank would.  Returns this rank's
        (samples, bytes, seconds); ``aggregate_over_ranks`` folds them."""
        keys, first, stride = shard_keys(list(keys), rank, world)
        kw.setdefault("first_index", first)
        kw.setdefault("index_stride", stride)
        f = h5lite.File(event_h5_file)

        def read(ks):
            def go():
                wins = []
                for k in ks:
                    ev = np.asarray(f[k])
                    if ev.ndim != 2 or ev.shape[1] != 4 or ev.dtype.itemsize != 4 or ev.dtype.kind not in "iu":
                        raise ValueError("%s[%r]: expected an (n, 4) int32 event array, got %s %s"
                                         % (event_h5_file, k, ev.shape, ev.dtype))
                    wins.append(np.ascontiguousarray(ev.view(np.int32)))
                return wins
            return go
        return self.run((read(keys[i:i + batch]) for i in range(0, len(keys), batch)), out_dir, **kw)


def shard_keys(keys, rank=None, world=None):
    """This rank's share of the sample keys, round-robin (``distributed.shard_indices``), with the numbering that keeps
    the reference's global ``counter`` (precompute_reps.py:440-466): -> (keys of this rank, first index, index stride)."""
    from .distributed import shard_indices
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if not 0 <= rank < world:
        raise ValueError("rank %d outside a world of %d" % (rank, world))
    return [keys[i] for i in shard_indices(len(keys), rank, world)], rank, world


def aggregate_over_ranks(samples, nbytes, seconds, device=None):
    """Every rank's (samples, bytes written, seconds) -> the job's figures on every rank: ONE all_gather of three float64
    scalars per rank.  Aggregate rate = total samples / the SLOWEST rank's seconds (the job is done when the last rank is)."""
    from .distributed import gather_rows
    rows = gather_rows([float(samples), float(nbytes), float(seconds)], device=device)
    n, b = float(rows[:, 0].sum()), float(rows[:, 1].sum())
    el = max(float(rows[:, 2].max()), 1e-12)
    per = (rows[:, 0] / rows[:, 2].clamp_min(1e-12)).tolist()
    return {"n_ranks": int(rows.shape[0]), "samples": int(n), "bytes": int(b), "seconds": el, "samples_per_s": n / el,
            "output_GBps": b / el / 1e9, "samples_per_rank": [int(v) for v in rows[:, 0].tolist()],
            "samples_per_s_per_rank_min": min(per), "samples_per_s_per_rank_max": max(per)}
