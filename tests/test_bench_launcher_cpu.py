"""bench.py --gpus N is launchable through the ONE entry point the driver uses.

`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself, rendezvous on 127.0.0.1,
run the barriers / MAX-over-ranks timing / the all-gather of GWD scalars, and print ONE JSON line with
n_gpus = 2.  Here the GPU legs are no-ops (EVREP_BENCH_DRYRUN=1, gloo backend); on the GPU box the same
code path runs on RCCL.  A world size that differs from --gpus is an error, never a silent 1-GPU run.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EVREP_BENCH_DRYRUN="1", **extra)
    return env


def _json_lines(out):
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]


def test_self_launch_two_ranks_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--gwd-pairs", "7"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout        # rank 0 only
    res = lines[0]
    assert res["n_gpus"] == 2 and res["dry_run"] is True and res["steps"] == 3 and res["scaling"] == "weak"
    g = res["gwd"]
    assert g["n_ranks"] == 2 and g["pairs"] == 7 and g["solves_per_rank"] == [4, 3] and g["all_solved"] is True
    assert "torch.distributed.run" in r.stderr      # it really re-launched itself
    # r06: a scaling run keeps config 3's sweep and config 5's precompute, per rank, folded by one all_gather each
    pr = res["per_rank"]
    sw, pre = pr["sweep_c3_per_rank"], pr["precompute"]
    assert sw["n_ranks"] == 2 and [row["builder"] for row in sw["rows"]] == ["time_surface_f64", "event_stack_f32", "tore_full_frame_f32"]
    assert all(len(row["build_ms_per_rank"]) == 2 and row["build_ms_min"] < row["build_ms_max"] for row in sw["rows"])
    assert pre["n_ranks"] == 2 and pre["samples"] == 256 and pre["samples_per_rank"] == [128, 128]
    assert abs(pre["seconds"] - 1.5) < 1e-9 and abs(pre["samples_per_s"] - 256 / 1.5) < 1e-6   # the slowest rank's clock


def test_single_rank_dry_run_prints_one_line():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "0", "--gwd-pairs", "3"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    (res,) = _json_lines(r.stdout)
    assert res["n_gpus"] == 1 and res["gwd"]["n_ranks"] == 1


def test_world_size_mismatch_is_an_error():
    # a launcher that started ONE rank for --gpus 2 must not produce a line labelled as anything
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and not _json_lines(r.stdout)
