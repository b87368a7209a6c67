#!/usr/bin/env python3
"""bench.py -- the hot path on MI355X: OptimizedRepresentation (ERGO-12) over a batch of windows.

One "step" = one pass of the hot path over one batch of synthetic event windows that is already
resident in HBM: the (y,x) binning pass + the ERGO-12 builder, device-resident events in,
device-resident (B, H, W, 12) float64 out (the reference's dtype, mixed_density_event_stack.py:36).
Workload = BASELINE.json configs[1]: 640x480, 12 channels, 50 000 events per window (the reference's
window size, gen1_2yolo.py:41), batch of 32 windows.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches it under ``python -m torch.distributed.run --nproc-per-node N``; run as a plain
``python bench.py --gpus N`` it re-launches ITSELF that way (one rank per GPU, rendezvous on 127.0.0.1), so
the one entry point always runs N ranks -- a world size that differs from --gpus is an error, never a silent
1-GPU run.  Every rank owns its own windows (weak scaling, no data-path collective); the only collective is
the all-gather of GWD scalars.  Rank 0 prints ONE JSON line.

``EVREP_BENCH_DRYRUN=1`` (tests/test_bench_launcher_cpu.py) keeps the launcher, the rendezvous, the barriers,
the MAX-over-ranks timing and the all-gather but runs them on the gloo backend with the GPU legs replaced by
no-ops, so the N > 1 control flow is exercised on a CPU-only box; the line it prints carries "dry_run": true
and is not a measurement.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver (N > 1)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, C = 480, 640, 12
EVENTS_PER_WINDOW = 50000
BATCH = 32
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)   # 0.2 s of load; boxes of the pool differ by up to 25 % in what they sustain, see DESIGN 5
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--events", type=int, default=EVENTS_PER_WINDOW)
    ap.add_argument("--out-dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gwd", action="store_true")
    ap.add_argument("--gwd-pairs", type=int, default=144,
                    help="GWD solves of the wall-time leg; 144 = the metric's 12x12 representation matrix")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc passes (~40 s) that measure roofline.traffic; the value is then "
                         "replayed from profiles/traffic.json and labelled so")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip the `sweep` leg (every builder at the reference's Gen1 shape and at configs 2 / 3, build-only "
                         "fraction of the HBM roof per builder; ~10 s)")
    ap.add_argument("--no-precompute", action="store_true",
                    help="skip the `precompute` leg (BASELINE config 5 on this GPU: 1280x720 windows -> (640,640,12) float32 "
                         "HDF5 files, end-to-end samples/s and GB/s beside the pinned D2H rate of the box; ~8 s)")
    ap.add_argument("--no-gw-extension", action="store_true",
                    help="skip the entropic Gromov-Wasserstein leg (extension, SURVEY 8 F5; ~1 s)")
    ap.add_argument("--probe-placement", type=int, default=0, metavar="N",
                    help="0 (default, r03): the output tensor is the FIRST allocation, as any caller gets it -- the builder paces its "
                         "stores (NOTES.md 3.2, Store pacing), so its launch no longer depends on where the tensor lies.  N > 1 (r02's default was "
                         "16): before the warm-up, allocate N candidate output tensors and keep the one the store probe runs fastest "
                         "into; the candidates' timings are printed in config.output_placement_probe.")
    ap.add_argument("--pacing", type=int, default=None, metavar="TICKS",
                    help="store pacing of the builder: default automatic; 0 = off (r02's behaviour); > 0 = hold in 10 ns ticks")
    ap.add_argument("--first-allocation", action="store_true",
                    help="after the timed region, also time 200 steps into the FIRST candidate allocation (what a run without the "
                         "placement probe gets) -> value_per_gpu_first_allocation; off by default so that a profiled run's kernel "
                         "statistics hold the measured steps only (all_us[0] of the probe already predicts it)")
    ap.add_argument("--resident-batches", type=int, default=12,
                    help="distinct resident batches the timed loop rotates over (12 x 25.6 MB of events > the 256 MB Infinity Cache)")
    ap.add_argument("--no-pipelined-value", action="store_true",
                    help="skip the second, labelled figure `pipelined` (bin k+1 beside build k, >= 200 steps after the timed region)")
    ap.add_argument("--pipeline", action="store_true",
                    help="overlap the binning pass of step k+1 with the builder of step k on a second HIP stream "
                         "(two resident batches alternate); default: bin + build back to back on one stream")
    return ap.parse_args()


def cpu_baseline(events_per_window, budget_s=12.0, threads_budget_s=6.0):
    """The oracle (C port of the reference's per-channel scatter passes): one host core for `budget_s`
    (the reported baseline), then every host core over independent windows for `threads_budget_s`
    ("single-threaded and with nproc threads over windows, core count printed")."""
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    from event_representation_study_amd.synthetic import make_events
    oracle.build()
    wins = [make_events(events_per_window, W, H, seed=1000 + i) for i in range(4)]
    oracle.ergo12(wins[0], H, W)  # warm
    t0 = time.perf_counter()
    done = 0
    while True:
        oracle.ergo12(wins[done % len(wins)], H, W)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    res = {"value": done * events_per_window / el, "unit": "events/s", "cores": 1, "kind": "port",
           "sample": "%d windows of %d events, 640x480x12 f64, oracle/evrep_oracle.c single thread, %.1f s"
                     % (done, events_per_window, el),
           "windows_per_s": done / el}
    ncpu = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else ncpu
    quota = None
    try:                                   # cgroup v2 CPU quota of the container: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    if ncpu > 1 and threads_budget_s > 0:
        def work(k):
            n = 0
            buf = np.empty((H, W, C), dtype=np.float64)        # one result buffer per thread, reused: first-touching a
            while time.perf_counter() < deadline:              # fresh 29.5 MB array per window would benchmark page faults
                oracle.ergo12(wins[(k + n) % len(wins)], H, W, out=buf)   # ctypes releases the GIL
                n += 1
            return n
        # wall-clock bounded whatever the scaling is; report the best of a few thread counts
        tried = []
        counts = sorted({min(ncpu, 8), min(ncpu, 32), ncpu})
        for nt in counts:
            t0 = time.perf_counter()
            deadline = t0 + threads_budget_s / len(counts)
            with ThreadPoolExecutor(nt) as ex:
                total = sum(ex.map(work, range(nt)))
            el2 = time.perf_counter() - t0
            tried.append({"value": total * events_per_window / el2, "unit": "events/s", "cores": nt,
                          "sample": "%d windows over %d threads, %.1f s" % (total, nt, el2)})
        res["all_cores"] = dict(max(tried, key=lambda r: r["value"]), host_cores=ncpu, affinity_cores=usable, cgroup_cpu_quota=quota,
                                tried=[(r["cores"], round(r["value"])) for r in tried])
    return res


def gwd_leg(rank, world, pairs, device, dry=False):
    """Wall time of `pairs` GWD solves at the reference's size (n ~ 12.5k events/quadrant, m = 14.4k
    representation points of C+2 = 14 features), sharded over ranks, scalars all-gathered."""
    sync = (lambda: None) if dry else torch.cuda.synchronize
    rng = np.random.default_rng(77)
    n, m = 12500, 14400
    mine = list(range(rank, pairs, world))
    costs = torch.zeros(pairs, dtype=torch.float64, device=device)
    if dry:
        def solve_mine():
            for i in mine:
                costs[i] = float(i + 1)
    else:
        from event_representation_study_amd.engine import gwd_padded_l1_batch
        Xs = torch.from_numpy(rng.random((n, 4))).to(device)
        Xt = torch.from_numpy(rng.random((m, 14)) * np.array([255.0] * 12 + [1.0, 1.0])).to(device)
        idx = torch.as_tensor(mine, dtype=torch.int64, device=device)
        nn = torch.full((len(mine),), n, dtype=torch.int64, device=device)
        mm = torch.full((len(mine),), m, dtype=torch.int64, device=device)
        zero = torch.zeros(len(mine), dtype=torch.int64, device=device)
        local = torch.zeros(len(mine), dtype=torch.float64, device=device)

        def solve_mine():
            # this rank's solves in ONE batched call (five launches for all of them, r03): every pair reads the same two
            # clouds here, as the metric's synthetic "12 x 12" matrix does; no host sync inside
            if mine:
                gwd_padded_l1_batch(Xs, nn, Xt, mm, n, m, xs_row=zero, xt_row=zero, out=local)
                costs[idx] = local
    for _ in range(2):     # warm: first launch of every kernel, scratch allocation
        solve_mine()
    costs.zero_()
    if world > 1:          # warm the collective as well (communicator set-up is not part of a solve)
        warm = [torch.zeros_like(costs) for _ in range(world)]
        torch.distributed.all_gather(warm, costs)
    sync()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    solve_mine()
    if world > 1:
        gathered = [torch.zeros_like(costs) for _ in range(world)]
        torch.distributed.all_gather(gathered, costs)
        costs = torch.stack(gathered).sum(0)
    sync()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    el = float(t.item())
    entries = pairs * (n * n + m * m)
    res = {"pairs": pairs, "n": n, "m": m, "wall_ms": el * 1e3, "kernel_entries_per_s": entries / el,
           "n_ranks": world, "solves_per_rank": [len(range(r, pairs, world)) for r in range(world)],
           "all_solved": bool((costs != 0).all().item()), "first_cost": float(costs[0].item())}
    if not dry:
        # matrix-core work of one solve (evrep_gwd.hip): upper-triangular 128 x 128 tiles of the L x L grid, each
        # 16 blocks of 32 x 32 pairs.  ALGORITHMIC work = the float32 form's: steps(d) v_mfma_f32_32x32x2_f32 of 4096 flop
        # per block and cloud present (steps = (d + 2) / 2 rounded up to 3 / 8 / 17: two extra inner dimensions carry the
        # squared norms).  EXECUTED since r03 for clouds of <= 15 dimensions: the same exponent matrix from exact
        # three-way bfloat16 splits, split_steps(d) v_mfma_f32_32x32x16_bf16 of 32768 flop (2 for d <= 4, else 6).
        split_steps = lambda d: 2 if 6 * d + 6 <= 32 else 6  # noqa: E731
        T = (max(n, m) + 127) // 128
        executed = 0
        for bi in range(T):
            for bj in range(bi, T):
                executed += 16 * 32768 * ((split_steps(4) if bj * 128 < n else 0) + (split_steps(14) if bj * 128 < m else 0))
        per_solve_s = el / max(len(mine), 1)          # this rank's solves: one batched call (6 launches in all)
        res["api"] = "evrep_gwd_padded_l1_batch"
        # The solve is bound by its exponentials, not by the matrix pipe ("bound by VALU/transcendental
        # throughput, not MFMA; report kernel entries/s"): one v_exp_f32 per kernel entry of the upper-triangular tiles,
        # a quarter-rate VALU instruction -- 4 lanes per clock and SIMD.  The bf16 matrix work the distances run on is
        # reported beside it against ITS peak, as executed (no "float32-equivalent" pricing).
        exp2_per_solve = 2 * sum(min(128 * (T - bi), 128 * T) * 128 for bi in range(T))
        cus, simds, lanes_per_clk, ghz = 256, 4, 4, 2.4
        exp_peak = cus * simds * lanes_per_clk * ghz * 1e9
        exps = exp2_per_solve / per_solve_s
        golden = None
        try:
            with open(os.path.join(ROOT, "tests", "golden", "gwd_fullsize.json")) as f:
                golden = json.load(f)["cost_f64"]      # the float64 oracle's value for exactly these clouds
        except Exception:
            pass
        if golden:
            res["first_cost_golden_f64"] = golden
            res["first_cost_rel_err"] = abs(res["first_cost"] - golden) / golden
        res["roofline"] = {"bound": "valu-transcendental",
                           "kernel": "k_gwd_tiles_split_batch<2, 6> (+ setup, statistics, scaling and final-sum launches, once per "
                                     "batch: the whole call is timed, so this is a lower bound of the tile kernel's own rate; its "
                                     "rocprofv3 average is in profiles/)",
                           "achieved": exps / 1e12, "peak": exp_peak / 1e12, "unit": "Texp2/s", "frac": exps / exp_peak,
                           "peak_is": "%d CU x %d SIMD x %d lanes/clk (v_exp_f32 is quarter rate) x %.1f GHz" % (cus, simds, lanes_per_clk, ghz),
                           "exp2_per_solve": exp2_per_solve, "us_per_solve": per_solve_s * 1e6,
                           "matrix_pipe": {"executed_bf16_mfma_flop_per_solve": executed,
                                           "executed_TFLOPs": executed / per_solve_s / 1e12, "bf16_matrix_peak_TFLOPs": 2500.0,
                                           "frac_of_bf16_peak": executed / per_solve_s / 1e12 / 2500.0,
                                           "note": "exponents from exact three-way bfloat16 splits of the float32 operands "
                                                   "(6 of 9 cross terms, dropped < 2^-23 relative): float32 MFMA chains and v_exp_f32 "
                                                   "exclude each other on a SIMD, bf16 MFMA does not"}}
    return res


def live_traffic(key):
    """HBM bytes per launch of the dominant kernel from the PMC counters, collected NOW: two rocprofv3 passes
    (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters only with --kernel-trace) over tools/pmc_workload.py,
    corrected by the calibration kernels that script runs first.  None if rocprofv3 is missing or a pass fails."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import parse_pmc
        tmp = tempfile.mkdtemp(prefix="evrep_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        found = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            subprocess.run([exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                            sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py")],
                           cwd="/tmp", env=env, check=True, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for dirpath, _, files in os.walk(d):
                for fn in files:
                    if fn.endswith("counter_collection.csv"):
                        found[ctr] = os.path.join(dirpath, fn)
        out = os.path.join(tmp, "traffic.json")
        with open(os.devnull, "w") as devnull:
            old = sys.stdout
            sys.stdout = devnull
            try:
                parse_pmc.main(found["FETCH_SIZE"], found["WRITE_SIZE"], out)
            finally:
                sys.stdout = old
        with open(out) as f:
            res = json.load(f)
        shutil.rmtree(tmp, ignore_errors=True)
        return res.get(key)
    except Exception as e:   # a profiler hiccup must not cost the bench line
        print("[bench] live PMC traffic unavailable (%s: %s); replaying profiles/traffic.json" % (type(e).__name__, e), file=sys.stderr)
        return None


def gw_extension_leg(device):
    """EXTENSION (no reference call computes it): entropic Gromov-Wasserstein at the reference's GWD
    problem size.  One outer iteration = the tensor product h1(C1) T h2(C2)^T as two MFMA GEMMs (float64:
    v_mfma_f64_16x16x4_f64) + the Sinkhorn passes; reported as TFLOP/s of the GEMM pair against the float64
    matrix-core peak, measured with HIP-side wall time over whole solves (so launches and the cheap init / plan
    kernels count against it)."""
    from event_representation_study_amd.gw_solver import entropic_gromov_wasserstein, flops_per_outer_iteration
    n, m = 12500, 14400
    g = torch.Generator(device=device).manual_seed(1)

    def kern(k, d):
        X = torch.rand((k, d), generator=g, device=device, dtype=torch.float64)
        D2 = torch.cdist(X, X) ** 2
        return torch.exp(-D2 / (2 * 0.49 * D2.mean() / 2))
    C1, C2 = kern(n, 4), kern(m, 14)

    def run(outer, sk):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, gw = entropic_gromov_wasserstein(C1, C2, None, None, "square_loss", 0.1, outer, sk, "f64", return_plan=False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, float(gw)
    run(1, 1)                                   # warm
    t1, _ = run(1, 1)                           # 2 GEMM pairs (iteration + final loss) + 1 Sinkhorn iteration
    t2, gw = run(1, 11)
    sk = (t2 - t1) / 10
    pair = (t1 - sk) / 2
    fl = flops_per_outer_iteration(n, m)
    return {"what": "entropic Gromov-Wasserstein (PGD + Sinkhorn), EXTENSION: not a reference computation, parity unpinned vs POT",
            "n": n, "m": m, "precision": "f64", "gw": gw, "gemm_pair_ms": pair * 1e3, "sinkhorn_iter_ms": sk * 1e3,
            "roofline": {"bound": "mfma", "kernel": "k_gw_gemm<double> x 2 (tensor product)", "achieved": fl / pair / 1e12,
                         "peak": 78.6, "unit": "TFLOP/s", "frac": fl / pair / 1e12 / 78.6, "flop_per_pair": fl},
            "sinkhorn_GBps": 2 * n * m * 8 / sk / 1e9}


def sweep_leg(device):
    """Every builder at the reference's real Gen1 shape (304x240, 50 000 events, gen1_2yolo.py:41-42,81-82) and at BASELINE
    configs 2 and 3: binning and build launch(es) timed separately with HIP events (tools/bench_sweep.py), the build against
    the same algorithmic-byte definition as the headline (16 B per event + the output tensor once).  Beside the uniform
    streams of SURVEY 8(d): the reference's moving-circle stream and an edge-cluster stream (synthetic.GENERATORS) for the
    three 12-channel builders, and the dense stress points N = 500 000 (640x480) / N = 1 000 000 (1280x720)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_sweep
    rows = bench_sweep.sweep(("gen1", "c2", "c3"), iters=10, device=str(device))
    three = ("optimized_f64", "event_stack_f32", "time_surface_f64")
    rows += bench_sweep.sweep(("gen1@circle", "gen1@edges", "c2@circle", "c2@edges"), iters=10, builders=three, device=str(device))
    rows += bench_sweep.sweep(("c3@circle", "c3@edges"), iters=10, builders=three + ("tore_full_frame_f32",), device=str(device))   # config 3 names TORE
    rows += bench_sweep.sweep(("c2-dense", "c3-1M"), iters=10, builders=three + ("tore_full_frame_f32", "voxel5_f64"), device=str(device))
    keep = ("config", "distribution", "builder", "binning_pass", "bin_ms", "build_ms", "build_GBps", "build_frac_of_8TBps",
            "events_per_s_bin_plus_build")
    # clustered rows: bin + build time over the uniform row of the same shape and builder
    base = {(r["config"], r["builder"]): r["bin_ms"] + r["build_ms"] for r in rows if r["distribution"] == "uniform"}
    out = []
    for r in rows:
        d = {k: r[k] for k in keep}
        if r["distribution"] != "uniform" and (r["config"], r["builder"]) in base:
            d["bin_plus_build_over_uniform"] = round((r["bin_ms"] + r["build_ms"]) / base[(r["config"], r["builder"])], 3)
        out.append(d)
    shapes = ("gen1", "c2", "c3", "c2-dense", "c3-1M")
    return {"roofline_bound": "hbm", "peak_GBps": HBM_PEAK_GBPS,
            "distributions": {"uniform": "x, y ~ U (SURVEY 8(d))",
                              "circle": "ev-licious generate_fake_events restated (fake_events.py:5-29), scaled to the frame",
                              "edges": "80 % of the events on 12 straight edges covering 5 % of the pixels"},
            "shapes": {k: dict(zip(("W", "H", "events_per_window", "batch"), bench_sweep.CONFIGS[k])) for k in shapes},
            "rows": out}


def precompute_leg(device, samples=512, batch=8, events=200000):
    """BASELINE config 5 on ONE GPU (tools/precompute_reps.py): a stream of 1280x720 windows -> bin + ERGO-12 + forced
    (640, 640) area resize on the GPU -> float32 D2H -> one HDF5 file per sample (dataset "repr", precompute_reps.py:432-435)
    in /dev/shm.  Beside it: the GPU side alone, and the pinned D2H rate of this box for one batch's 157 MB."""
    import shutil
    import tempfile
    from event_representation_study_amd.precompute import RepPrecomputer
    from event_representation_study_amd.synthetic import make_events
    Hc, Wc = 720, 1280
    pool = [make_events(events, Wc, Hc, seed=9000 + i) for i in range(batch)]
    out_dir = tempfile.mkdtemp(prefix="evrep_c5_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        pc = RepPrecomputer(Hc, Wc, 640, "optimized", device=str(device), writers=12, loaders=3)
        pc.reserve([(batch, 640, 640, 12)])                                               # the pinned ring, up front
        pc.run([pool, pool], out_dir, keep_files=False)                                   # warm-up: tap tables, first launches
        n, nbytes, el = pc.run(([pool[i % batch] for i in range(batch)] for _ in range(samples // batch)), out_dir,
                               keep_files=False)
        for _ in range(2):
            rep = pc.represent(pool)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            rep = pc.represent(pool)
        torch.cuda.synchronize()
        gpu_s = (time.perf_counter() - t0) / (10 * batch)
        host = torch.empty(tuple(rep.shape), dtype=torch.float32, pin_memory=True)
        host.copy_(rep, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            host.copy_(rep, non_blocking=True)
        torch.cuda.synchronize()
        d2h = 10 * host.numel() * 4 / (time.perf_counter() - t0) / 1e9
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    per = 640 * 640 * 12 * 4
    return {"workload": "%d windows of %d events, 1280x720 -> (640,640,12) float32, one HDF5 file per sample in /dev/shm" % (n, events),
            "samples_per_s": n / el, "output_GBps": nbytes / el / 1e9, "seconds": el,
            "gpu_only_us_per_sample": gpu_s * 1e6, "gpu_only_samples_per_s": 1.0 / gpu_s,
            "pinned_d2h_GBps": d2h, "pcie_gen5_x16_GBps": 63.0, "d2h_bound_samples_per_s": d2h * 1e9 / per,
            "bytes_per_sample_over_pcie": per}


def host_cores():
    """Cores this process may use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def per_rank_legs(rank, world, device, dry, samples=256, batch=8, events=200000):
    """N > 1 (VERDICT r05 missing 1): every rank runs BASELINE config 3's sweep (TimeSurface + EventStack + TORE
    on 1280x720 windows) and config 5's precompute (its share of the sample stream -> one HDF5 file per sample under the
    GLOBAL sample number, as precompute_reps.py:439-466's Pool(8) leaves them) on its own GPU; the figures meet in ONE
    all_gather of a few scalars per leg (distributed.gather_rows) -- no data-path collective.  With EVREP_BENCH_DRYRUN=1 the
    GPU work is replaced by stand-in figures and the exchange runs on gloo."""
    import shutil
    import tempfile
    from event_representation_study_amd.distributed import gather_rows
    names = ("time_surface_f64", "event_stack_f32", "tore_full_frame_f32")
    if dry:
        local = [1.0 + rank, 0.1 * (rank + 1)] + [0.2 * (rank + 1)] * len(names) + [1e9] * len(names)
    else:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_sweep
        rows = {r["builder"]: r for r in bench_sweep.sweep(("c3",), iters=10, builders=names, device=str(device))}
        local = [float(rank), rows[names[0]]["bin_ms"]] + [rows[n]["build_ms"] for n in names] \
            + [rows[n]["algorithmic_bytes"] for n in names]
    g = gather_rows(local, device=device)
    Wc, Hc, N3, B3 = 1280, 720, 200000, 8
    sweep = {"workload": "config 3 on every rank: %d windows of %d events, %dx%d; per-rank HIP-event timings" % (B3, N3, Wc, Hc),
             "n_ranks": int(g.shape[0]), "bin_ms_min": float(g[:, 1].min()), "bin_ms_max": float(g[:, 1].max()), "rows": []}
    for j, n in enumerate(names):
        ms, alg = g[:, 2 + j], g[:, 2 + len(names) + j]
        sweep["rows"].append({"builder": n, "build_ms_min": float(ms.min()), "build_ms_max": float(ms.max()),
                              "build_ms_per_rank": [round(float(v), 4) for v in ms.tolist()],
                              "build_frac_of_8TBps_min": float((alg / ms / 1e6 / HBM_PEAK_GBPS).min()),
                              "events_per_s_all_ranks_bin_plus_build":
                                  float((B3 * N3 / ((g[:, 1] + ms) * 1e-3)).sum())})
    # config 5: the sample stream dealt round-robin to the ranks (sample i -> rank i mod N, written as <i>.h5)
    from event_representation_study_amd.precompute import aggregate_over_ranks, shard_keys
    mine, first, stride = shard_keys(list(range(samples)), rank, world)
    if dry:
        trip = (len(mine), len(mine) * 640 * 640 * 12 * 4, 1.0 + 0.5 * rank)
    else:
        from event_representation_study_amd.precompute import RepPrecomputer
        from event_representation_study_amd.synthetic import make_events
        pool = [make_events(events, Wc, Hc, seed=9000 + rank * 100 + i) for i in range(batch)]
        out_dir = tempfile.mkdtemp(prefix="evrep_c5_r%d_" % rank, dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            pc = RepPrecomputer(Hc, Wc, 640, "optimized", device=str(device), writers=max(2, min(12, host_cores() // world)),
                                loaders=2)
            pc.reserve([(batch, 640, 640, 12)])
            pc.run([pool, pool], out_dir, keep_files=False)
            if world > 1:
                torch.distributed.barrier()           # the ranks start their shares together
            stream = ([pool[i % batch] for i in range(len(mine[k:k + batch]))] for k in range(0, len(mine), batch))
            trip = pc.run(stream, out_dir, keep_files=False, first_index=first, index_stride=stride)
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    pre = aggregate_over_ranks(*trip, device=device)
    pre["workload"] = ("%d samples of %d events dealt round-robin to %d ranks, 1280x720 -> (640,640,12) float32, one HDF5 file "
                       "per sample under its global number in /dev/shm" % (samples, events, world))
    pre["host_cores_shared_by_all_ranks"] = host_cores()
    return {"sweep_c3_per_rank": sweep, "precompute": pre}


def achievable_rates(device, nbytes=1 << 30, iters=10):
    """What this box's HBM delivers to the simplest kernels, measured in this process: a 1 GiB fill (write-only, the
    builder's own traffic pattern) and a 1 GiB device-to-device copy (read + write counted).  The 8 TB/s of `peak` is the
    HBM3E specification; `frac_of_achievable` prices the builder against the better of these two."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    b = torch.empty_like(a)
    out = {}
    for name, fn, moved in (("fill", lambda: a.fill_(1.0), nbytes), ("copy", lambda: b.copy_(a), 2 * nbytes)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out["measured_%s_GBps" % name] = moved * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9
    out["achievable_GBps"] = max(out["measured_fill_GBps"], out["measured_copy_GBps"])
    del a, b
    return out


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: run N ranks of this file under
    torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and hand back its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=dict(os.environ, EVREP_BENCH_SELF_LAUNCHED="1"))


def main():
    args = parse()
    dry = os.environ.get("EVREP_BENCH_DRYRUN") == "1"
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        if torch.cuda.device_count() < args.gpus and not launched:
            raise SystemExit("--gpus %d but only %d device(s) visible" % (args.gpus, torch.cuda.device_count()))
    if args.gpus > 1 and not launched:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:   # never a silently smaller (or larger) run than the one asked for
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if dry:
        device = torch.device("cpu")
        sync = lambda: None  # noqa: E731
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)

    B, N = args.batch, args.events
    dtype = torch.float64 if args.out_dtype == "f64" else torch.float32
    # --pipeline: two resident batches (different windows) alternate, as a stream of batches would, and the
    # binning pass of step k+1 overlaps the builder of step k on a second HIP stream (measured: ~5 % more
    # throughput, but the builder's own launch time is inflated by the sharing, so it is not the default)
    # (r04) the timed loop rotates over `nbatch` DISTINCT resident batches: 12 x 25.6 MB of events (+ 12 workspaces) is more
    # than the 256 MB Infinity Cache holds, so the read side of a step comes from HBM, as it does for a stream of batches
    nbatch = 2 if args.pipeline else max(1, args.resident_batches)
    nbuf = 2 if args.pipeline else 1
    pipe = None
    if dry:
        def step(k, ev_pair=None):
            pass
    else:
        from event_representation_study_amd.engine import BinBuildPipeline, EventBatch
        from event_representation_study_amd.synthetic import make_events
        batches, outs = [], []
        for j in range(nbatch):
            wins = [make_events(N, W, H, seed=rank * 100000 + j * B + i) for i in range(B)]  # seed = window index
            batches.append(EventBatch.from_numpy(wins, H, W, device=device))
            if args.pacing is not None:
                import ctypes
                from event_representation_study_amd._lib import check
                check(batches[-1].lib.evrep_plan_set_pacing(ctypes.byref(batches[-1].plan), args.pacing), "evrep_plan_set_pacing")
            if j >= nbuf:
                continue            # one output tensor: it is written, never read, by the step
            if args.probe_placement > 1:
                from event_representation_study_amd.engine import probe_output_placement
                # timed with the library's store probe (the builder's write footprint, no builder launch: the k_mdes
                # statistics of a profiled run hold the real steps only)
                try:
                    # rounds of N candidates until one takes the probe at >= 6.5 TB/s (a fast region), at most 4 rounds
                    o, best_us, all_us, first_alloc = probe_output_placement((B, H, W, C), dtype, candidates=args.probe_placement,
                                                                             device=device, keep_first=True, good_GBps=6500.0,
                                                                             max_candidates=4 * args.probe_placement)
                    placement = {"candidates": len(all_us), "writer": "evrep_probe_store", "best_us": round(best_us, 1),
                                 "all_us": [round(x, 1) for x in all_us]}
                except torch.OutOfMemoryError:   # a crowded device: take one allocation, say so
                    torch.cuda.empty_cache()
                    o = torch.empty((B, H, W, C), dtype=dtype, device=device)
                    placement = {"candidates": 1, "note": "out of memory while probing candidates"}
                outs.append(o)
            else:
                outs.append(torch.empty((B, H, W, C), dtype=dtype, device=device))
        pipe = BinBuildPipeline(device) if args.pipeline else None

        def build(k, ev_pair):
            def fn(batch):
                if ev_pair is not None:
                    ev_pair[0].record()
                batch.optimized(scale=1.0, dtype=dtype, out=outs[k % nbuf])
                if ev_pair is not None:
                    ev_pair[1].record()
            return fn

        def step(k, ev_pair=None):
            batch = batches[k % nbatch]
            if pipe is None:
                if ev_pair is not None:
                    ev_pair[2].record()
                batch.rebin()
                build(k, ev_pair)(batch)
            else:
                pipe.submit(batch, build(k, ev_pair))

    if not dry and pipe is None:
        # initialisation, not warm-up: every resident batch is binned and built once, so that its workspace pages exist and
        # are mapped before the W warm-up steps (a short run -- the driver's W = 5 -- would otherwise time first touches)
        # ... and the device's clocks are up: the pass is repeated until >= 150 ms of back-to-back steps have run (r05: the
        # driver's 5 + 20 steps are 4 ms of GPU work -- on a cold device they were timed on the clock ramp)
        t_init = time.perf_counter()
        init_steps = 0
        while init_steps < nbatch or time.perf_counter() - t_init < 0.15:
            for j in range(nbatch):
                step(j)
            init_steps += nbatch
            sync()
        init_ms = (time.perf_counter() - t_init) * 1e3
    for k in range(args.warmup):
        step(k)
    if pipe is not None:
        pipe.drain()
    placement_note = locals().get("placement")
    # HIP events bracket the builder launch on every `ev_stride`-th timed step (each record is a marker packet in the
    # stream: bracketing every launch costs ~4 % of the step; every 4th still gives K/4 live samples inside the timed region)
    ev_stride = 4 if args.steps >= 8 else 1
    pairs = None if dry else {k: tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
                              for k in range(0, args.steps, ev_stride)}
    sync()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, None if dry else pairs.get(k))
    if pipe is not None:
        pipe.drain()
    sync()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=device)
    per_rank_ms = [el / max(args.steps, 1) * 1e3]
    if world > 1:
        every = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(every, t)
        per_rank_ms = [float(x.item()) / max(args.steps, 1) * 1e3 for x in every]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    el = max(float(t.item()), 1e-9)

    elem = 8 if dtype == torch.float64 else 4
    alg_bytes = B * (16 * N + elem * H * W * C)  # read every event once, write every output once

    result = {
        "metric": "events/sec/GPU (OptimizedRep, Gen1 640x480) + 12-rep GWD matrix wall-time",
        "value": world * B * N * args.steps / el,
        "unit": "events/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if dtype == torch.float64 else "f32",
        "data": "synthetic",
        "config": {"workload": "OptimizedRepresentation (ERGO-12) 640x480x12, %d events/window, batch %d windows/GPU, "
                               "bin + build per step%s, events and output resident in HBM"
                               % (N, B, "" if not pipe else " (bin of step k+1 overlapped with build of step k on a second stream)"),
                   "pipeline": pipe is not None,
                   "untimed_initialisation": "every resident batch binned and built, repeated for %.0f ms (%d steps) before the "
                                             "W warm-up steps: pages mapped, clocks ramped" % (locals().get("init_ms", 0.0), locals().get("init_steps", 0)),
                   "events_per_window": N, "batch": B, "height": H, "width": W, "channels": C,
                   "parallelism": "windows sharded over %d GPU(s), one process per GPU, no data-path collective" % world},
        "ms_per_step_per_rank": per_rank_ms,
        "windows_per_s": world * B * args.steps / el,
        "algorithmic_GBps_whole_step": world * alg_bytes * args.steps / el / 1e9,
        # third-party boundaries of the reference that are restated, never run (packages absent from the image and from
        # /root/reference): what the parity claims of this line do NOT pin
        "parity_unpinned": ["torch_scatter.scatter", "POT", "tonic", "cv2"],
    }
    if placement_note:
        # the same step into the FIRST candidate allocation (what a run without the probe gets), 200 steps, beside the line's value
        first = locals().get("first_alloc")
        if first is not None and not args.pipeline and args.first_allocation:
            keep, outs[0] = outs[0], first
            for k in range(20):
                step(0)
            sync()
            t1 = time.perf_counter()
            for k in range(200):
                step(0)
            sync()
            placement_note["first_allocation_ms_per_step"] = (time.perf_counter() - t1) / 200 * 1e3
            outs[0] = keep
            # beside `value`: this rank's rate had the output tensor been the first allocation (no probe)
            result["value_per_gpu_first_allocation"] = B * N / (placement_note["first_allocation_ms_per_step"] * 1e-3)
        result["config"]["output_placement_probe"] = placement_note   # --probe-placement 0 takes the first allocation instead
    if dry:
        result["dry_run"] = True      # launcher / rendezvous / collectives only: NOT a measurement
        result["value"] = 0.0
    else:
        builder_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in pairs.values()]))  # the builder (main + hot launch), HIP events on its stream
        bin_ms = float(np.mean([c.elapsed_time(a) for a, _, c in pairs.values()])) if pipe is None else None
        achieved = alg_bytes / (builder_ms * 1e-3) / 1e9
        result["roofline"] = {"bound": "hbm", "kernel": "k_mdes<%s>" % ("double" if elem == 8 else "float"),
                              "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "avg_launch_ms": builder_ms,
                              "algorithmic_bytes_per_launch": alg_bytes}
        # per-kernel microseconds of the step, HIP events inside the timed region (every 4th step)
        result["step_us"] = {"bin (k_block_keysort)": None if bin_ms is None else bin_ms * 1e3,
                             "build (k_mdes main launch + hot launch)": builder_ms * 1e3,
                             "resident_batches_rotated": nbatch,
                             "resident_event_MB": nbatch * B * N * 16 / 1e6}
        ach = achievable_rates(device)
        result["roofline"].update(ach)
        result["roofline"]["frac_of_achievable"] = achieved / ach["achievable_GBps"]
        tr = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr):
            try:
                with open(tr) as f:
                    trj = json.load(f)
                key = "k_mdes_f64" if elem == 8 else "k_mdes_f32"
                if key in trj and trj[key].get("batch") == B and trj[key].get("events") == N:
                    result["roofline"]["traffic"] = trj[key]["hbm_bytes_per_launch"]
                    result["roofline"]["traffic_source"] = "replayed from profiles/traffic.json (%s); not collected in this run" \
                        % trj[key].get("source")
            except Exception:
                pass
    if not dry and pipe is None and world == 1 and not args.no_pipelined_value:
        # the overlapped counterpart of `value`: the same steps over the same resident batches with
        # the binning pass of step k+1 on a second HIP stream beside the builder of step k.  A second, LABELLED figure:
        # `value` stays the back-to-back form on one stream.
        from event_representation_study_amd.engine import BinBuildPipeline as _BBP
        p2 = _BBP(device)
        ksteps = max(200, args.steps)

        def pstep(k):
            p2.submit(batches[k % nbatch], lambda b: b.optimized(scale=1.0, dtype=dtype, out=outs[0]))
        for k in range(24):
            pstep(k)
        p2.drain()
        sync()
        t1 = time.perf_counter()
        for k in range(ksteps):
            pstep(k)
        p2.drain()
        sync()
        pel = time.perf_counter() - t1
        result["pipelined"] = {"what": "bin of step k+1 on a second HIP stream beside the build of step k (engine.BinBuildPipeline), "
                                       "same resident batches and output tensor; NOT `value`",
                               "steps": ksteps, "ms_per_step": pel / ksteps * 1e3, "events_per_s": B * N * ksteps / pel,
                               "algorithmic_GBps_whole_step": alg_bytes * ksteps / pel / 1e9,
                               "whole_step_frac_of_8TBps": alg_bytes * ksteps / pel / 1e9 / HBM_PEAK_GBPS,
                               "over_value": (B * N * ksteps / pel) / result["value"]}
    if not args.no_gwd:   # while the GPU is still warm: the CPU baseline below idles it for ~20 s
        result["gwd"] = gwd_leg(rank, world, args.gwd_pairs, device, dry)
    if world > 1 and not (args.no_sweep and args.no_precompute):
        # every rank, its own GPU: config 3's sweep + its share of config 5 (a scaling run measures every config, r06)
        result["per_rank"] = per_rank_legs(rank, world, device, dry)
    if rank == 0 and world == 1 and not dry and not args.no_sweep:
        result["sweep"] = sweep_leg(device)
    if rank == 0 and world == 1 and not dry and not args.no_precompute:
        try:
            result["precompute"] = precompute_leg(device)
        except Exception as e:      # a full /dev/shm must not cost the bench line
            result["precompute"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not dry and not args.no_gw_extension:   # single-GPU leg: not part of a scaling run
        result["gw_extension"] = gw_extension_leg(device)
    if not dry and rank == 0 and world == 1 and not args.no_live_traffic and B == BATCH and N == EVENTS_PER_WINDOW:
        live = live_traffic("k_mdes_f64" if elem == 8 else "k_mdes_f32")   # last GPU leg: two profiler passes, ~40 s
        if live is not None:
            result["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
            result["roofline"]["traffic_read_bytes"] = live["read_bytes"]
            result["roofline"]["traffic_write_bytes"] = live["write_bytes"]
            result["roofline"]["traffic_source"] = ("collected in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                                                    "(two separate passes of tools/pmc_workload.py), calibrated on a 1 GiB fill "
                                                    "and a 1 GiB copy of the same process as MI355X_MICROARCH.md prescribes")
    if rank == 0 and not args.no_cpu_baseline and not dry:
        # rank 0's host cores, after the timed region and every GPU leg (the other ranks wait at the barrier below)
        result["cpu_baseline"] = cpu_baseline(N)
    elif rank == 0:
        result["cpu_baseline"] = None
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
