"""Config 5 on N ranks (reference: Pool(8) of CPU workers, precompute_reps.py:439-466): the sample keys are dealt
round-robin, every rank writes its samples under their GLOBAL numbers, the figures meet in one all_gather.  Here on gloo,
world size 2, with the GPU stage of RepPrecomputer replaced by a stand-in that writes the events' checksum (the host
logic -- sharding, numbering, aggregation -- is what is under test; the GPU stage has its own -m gpu tests)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5 = "events.h5"      # stand-in container (below): seven distinct (n, 4) int32 datasets, as precompute_reps.py:408-409 reads them
KEYS = ["s%d" % i for i in range(7)]


class _Container(dict):
    """What run_h5 needs of h5lite.File: key -> (n, 4) int32 array."""
    def __init__(self, path):
        rng = np.random.default_rng(5)
        super().__init__({k: rng.integers(0, 1000, size=(10 + 3 * i, 4)).astype(np.int32) for i, k in enumerate(KEYS)})


def _patched_precompute():
    import types
    from event_representation_study_amd import h5lite, precompute
    precompute.h5lite = types.SimpleNamespace(File=_Container, write_dataset_file=h5lite.write_dataset_file)
    return precompute


def test_shard_keys_partition_and_numbering():
    from event_representation_study_amd.precompute import shard_keys
    keys = ["k%d" % i for i in range(11)]
    for world in (1, 2, 3, 8, 16):
        seen = {}
        for rank in range(world):
            mine, first, stride = shard_keys(keys, rank, world)
            for j, k in enumerate(mine):
                seen[first + j * stride] = k
        assert [seen[i] for i in range(len(keys))] == keys        # every sample once, under its own global number
    import pytest
    with pytest.raises(ValueError):
        shard_keys(keys, 2, 2)


def _fake_run(self, window_batches, out_dir, keep_files=True, first_index=0, index_stride=1):
    """RepPrecomputer.run's contract without a GPU: one file per sample, numbered first_index + k * index_stride."""
    import time
    from event_representation_study_amd import h5lite
    os.makedirs(out_dir, exist_ok=True)
    t0, k, nbytes = time.perf_counter(), 0, 0
    for item in window_batches:
        for w in (item() if callable(item) else item):
            arr = np.asarray([w.shape[0], int(w.astype(np.int64).sum())], dtype=np.float32)
            h5lite.write_dataset_file(os.path.join(out_dir, "%d.h5" % (first_index + k * index_stride)), "repr", arr)
            nbytes += arr.nbytes
            k += 1
    return k, nbytes, time.perf_counter() - t0


def _worker(rank, world, port, out_dir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    precompute = _patched_precompute()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pc = precompute.RepPrecomputer.__new__(precompute.RepPrecomputer)
    pc.run = _fake_run.__get__(pc)
    trip = pc.run_h5(H5, KEYS, out_dir, batch=2)             # rank / world from the initialised group
    agg = precompute.aggregate_over_ranks(*trip)
    q.put((rank, trip[0], agg))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_precompute_leaves_the_files_one_rank_would(tmp_path):
    import torch.multiprocessing as mp
    from event_representation_study_amd import h5lite
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 23500 + (os.getpid() % 2000)
    two = str(tmp_path / "two")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, two, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in procs)
    assert [r[1] for r in res] == [4, 3]                      # 7 keys dealt round-robin
    agg = res[0][2]
    assert agg == res[1][2]                                   # every rank ends with the same figures
    assert agg["n_ranks"] == 2 and agg["samples"] == 7 and agg["samples_per_rank"] == [4, 3] and agg["bytes"] == 7 * 8
    assert agg["samples_per_s"] <= agg["samples"] / max(1e-12, agg["seconds"]) * (1 + 1e-9)
    # one rank alone leaves exactly the same files
    precompute = _patched_precompute()
    pc = precompute.RepPrecomputer.__new__(precompute.RepPrecomputer)
    pc.run = _fake_run.__get__(pc)
    one = str(tmp_path / "one")
    n, _, _ = pc.run_h5(H5, KEYS, one, batch=3, rank=0, world=1)
    assert n == 7 and sorted(os.listdir(one)) == sorted(os.listdir(two)) == sorted("%d.h5" % i for i in range(7))
    for i in range(7):
        a = h5lite.File(os.path.join(one, "%d.h5" % i))["repr"][()]
        b = h5lite.File(os.path.join(two, "%d.h5" % i))["repr"][()]
        np.testing.assert_array_equal(a, b)
