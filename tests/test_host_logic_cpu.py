"""CPU-only: host-side logic around the C ABI (synthetic streams, field conversion, the GWD quadrant
harness, dispatcher table, sharding helpers incl. a world_size-2 gloo run)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

from event_representation_study_amd.synthetic import from_structured, make_events, to_structured


def test_synthetic_streams_are_seeded_and_sorted():
    a = make_events(5000, 64, 48, seed=3)
    b = make_events(5000, 64, 48, seed=3)
    assert np.array_equal(a, b) and a.dtype == np.int32 and a.shape == (5000, 4)
    assert np.all(np.diff(a[:, 2]) >= 0) and a[0, 2] == 0
    assert set(np.unique(a[:, 3])) == {-1, 1}
    assert set(np.unique(make_events(500, 8, 8, seed=1, polarity="01")[:, 3])) == {0, 1}
    d = make_events(100, 8, 8, seed=1, dup_last=7)
    assert np.all(d[-7:, 2] == d[-1, 2])
    g = load_golden("c1_304x240_n10000_pm1")        # the fixtures were generated from the same generator
    assert np.array_equal(make_events(10000, 304, 240, seed=101), g["events"])


def test_structured_round_trip_and_float_fields():
    ev = make_events(100, 16, 12, seed=2)
    rec = to_structured(ev)
    assert rec.dtype.names == ("x", "y", "t", "p") and all(rec.dtype[n] == np.dtype("<i4") for n in rec.dtype.names)
    assert np.array_equal(from_structured(rec), ev)
    f8 = np.empty(100, dtype=[("x", "<f8"), ("y", "<f8"), ("t", "<f8"), ("p", "<f8")])   # n_imagenet's layout
    for k, n in enumerate(("x", "y", "t", "p")):
        f8[n] = ev[:, k]
    assert np.array_equal(from_structured(f8), ev)
    f8["t"][3] += 0.5
    f8["x"][5] += 0.25
    with pytest.raises(NotImplementedError):
        from_structured(f8)                               # builders whose reference does NOT truncate stay strict
    cut = from_structured(f8, truncate=True)              # MDES / EventStack: the reference's astype truncation
    assert np.array_equal(cut, ev)
    f8["t"] += 5e9                                        # absolute microseconds beyond int32
    with pytest.raises(OverflowError):
        from_structured(f8, truncate=True)
    reb = from_structured(f8, truncate=True, rebase_t=True)
    assert np.array_equal(reb[:, 2], ev[:, 2] - ev[:, 2].min()) and np.array_equal(reb[:, [0, 1, 3]], ev[:, [0, 1, 3]])
    i8 = np.empty(4, dtype=[("x", "<i4"), ("y", "<i4"), ("t", "<i8"), ("p", "<i4")])
    i8["x"], i8["y"], i8["p"], i8["t"] = 1, 2, 1, [2**31, 2**31 + 1, 2**31 + 2, 2**31 + 5]
    with pytest.raises(OverflowError):                    # no silent wrap of timestamps >= 2^31
        from_structured(i8)
    assert list(from_structured(i8, rebase_t=True)[:, 2]) == [0, 1, 2, 5]


def test_float_field_goldens_through_host_narrowing(oracle):
    """The reference's outputs on all-'<f8' fields (boundary.npz) = the oracle on the host-narrowed events."""
    from conftest import assert_bit_equal
    g = load_golden("boundary")
    rec = np.empty(g["float_rec_x"].shape[0], dtype=[("x", "<f8"), ("y", "<f8"), ("t", "<f8"), ("p", "<f8")])
    for n in "xytp":
        rec[n] = g["float_rec_" + n]
    H, W = int(g["float_H"]), int(g["float_W"])
    ev = from_structured(rec, truncate=True, rebase_t=True)
    assert_bit_equal(oracle.ergo12(ev, H, W), g["float_ergo12"], "float ergo12")
    assert_bit_equal(oracle.mdes(ev, H, W, [0, 3, 5, 1], ["timestamp", "count_neg", "polarity", "timestamp_pos"],
                                 ["mean", "sum", "variance", "max"]), g["float_mdes"], "float mdes")
    r2p = (rec["p"] + 1) // 2
    assert_bit_equal(oracle.event_stack_split(rec["x"], rec["y"], r2p, rec["t"], rec["t"][-1], H, W),
                     g["float_event_stack"], "float event stack")


def test_otmi_point_clouds_match_oracle_restatement(oracle):
    from event_representation_study_amd.representations.representation_search.compute_otmi import otmi_point_clouds
    g = load_golden("gwd")
    ours = otmi_point_clouds(torch.from_numpy(g["otmi_events"].copy()), g["otmi_rep"], int(g["otmi_H"]),
                             int(g["otmi_W"]), int(g["otmi_S"]))
    ref = oracle.otmi_point_clouds(g["otmi_events"], g["otmi_rep"], int(g["otmi_H"]), int(g["otmi_W"]),
                                   int(g["otmi_S"]))
    assert len(ours) == len(ref) == 3
    for (a, b), (c, d) in zip(ours, ref):
        assert a.dtype == np.float32 and b.dtype == np.float64
        assert np.array_equal(a, c) and np.array_equal(b, d)
    # and the oracle's closed form on those clouds reproduces the reference's harness output
    cost = np.mean([oracle.gwd(a, b) for a, b in ours])
    assert abs(cost - float(g["otmi_cost"])) <= 1e-5 * float(g["otmi_cost"])


def test_dispatcher_branch_order():
    from event_representation_study_amd.representations import gen1_transforms as g1
    names = [n for n, _, _ in g1._BRANCHES]
    assert names.index("MixedDensityEventStack") < names.index("EventStack")   # gen1_transforms.py:27 before :33
    assert names == ["ToVoxelGrid", "MixedDensityEventStack", "EventStack", "ToImage", "TORE", "ToTimesurface"]
    with pytest.raises(UnboundLocalError):
        g1.get_item_transform(None, "nothing", None, 1, 1, 0, 0)


def test_shard_indices_cover_everything_once():
    from event_representation_study_amd.distributed import shard_indices
    for n, w in ((0, 1), (7, 2), (36, 8), (5, 8)):
        seen = sorted(i for r in range(w) for i in shard_indices(n, r, w))
        assert seen == list(range(n))


def _gloo_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from event_representation_study_amd.distributed import gather_scalars, mean_cp, shard_indices, sharded_scores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = list(range(9))                                  # 3 samples x 3 quadrants
    mine = shard_indices(len(items))
    vec = gather_scalars({i: 0.1 * (i + 1) for i in mine}, len(items))
    vec2 = sharded_scores(lambda it: float(it) ** 2, items)
    q.put((rank, mine, vec.tolist(), vec2.tolist(), mean_cp(vec)))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_gather_of_gwd_scalars():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, v0, w0, cp0), (r1, mine1, v1, w1, cp1) = res
    assert mine0 == [0, 2, 4, 6, 8] and mine1 == [1, 3, 5, 7]
    want = [0.1 * (i + 1) for i in range(9)]
    assert np.allclose(v0, want) and v0 == v1                # every rank ends with the full vector
    assert w0 == w1 == [float(i) ** 2 for i in range(9)]
    assert abs(cp0 - np.mean(want)) < 1e-12 and cp0 == cp1


def test_precompute_parts_of_one_batch_never_share_pinned_storage(monkeypatch):
    """TORE hands one part per sample when the bounding-box frames differ; more same-shape parts than the
    ring is deep used to receive the same host buffer inside one batch.  Every batch now cuts its parts out of ONE flat
    buffer of the ring."""
    import torch
    from event_representation_study_amd import precompute

    real_empty = torch.empty
    monkeypatch.setattr(precompute.torch, "empty", lambda *a, pin_memory=False, **k: real_empty(*a, **k))
    pc = precompute.RepPrecomputer.__new__(precompute.RepPrecomputer)
    import threading
    pc.nwriters, pc._ring, pc._ring_lock = 4, {}, threading.Lock()
    shapes = [(37, 640, 12)] * 25 + [(640, 11, 12)] * 3          # more same-shape parts than a ring would be deep
    held = []
    for _ in range(12):
        views, token = pc._pinned_parts(shapes)
        assert [tuple(v.shape) for v in views] == shapes
        spans = sorted((v.data_ptr(), v.data_ptr() + v.numel() * 4) for v in views)
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "parts of one batch overlap"
        held.append((views[0].untyped_storage().data_ptr(), token))
    # a buffer is never handed out while its writer still holds it -- whatever order the writers finish in
    assert len({p for p, _ in held}) == 12
    pc._release(held[7][1])                                       # the EIGHTH writer finishes first
    views, token = pc._pinned_parts(shapes)
    assert views[0].untyped_storage().data_ptr() == held[7][0]    # its buffer, and only its buffer, is recycled
    views2, token2 = pc._pinned_parts(shapes)
    assert views2[0].untyped_storage().data_ptr() not in {p for p, _ in held}   # everything else is busy: a new one
    for _, t in held:
        pc._release(t)
    assert len(pc._ring["flat"]) == 13


def test_result_slot_is_free_only_when_nothing_outside_refers_to_it():
    """The pinned result pool of the per-sample wrappers (representations/_common.py): a buffer is handed out again only when
    no array handed out earlier -- or any view of one -- is still alive."""
    import torch
    from event_representation_study_amd.representations._common import _ResultSlot
    slot = _ResultSlot((4, 5, 3), torch.float64, pin=False)
    assert slot.free()
    a = slot.master.view()
    assert not slot.free()
    b = a[..., 1]                      # a view of a view still has the master as its base
    c = a.reshape(-1)[:6].reshape(2, 3)
    del a
    assert not slot.free()
    t = torch.from_numpy(b)            # torch.tensor()-style consumers keep it alive as well
    del b, c
    assert not slot.free()
    del t
    assert slot.free()
