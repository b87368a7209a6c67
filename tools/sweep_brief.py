#!/usr/bin/env python3
"""bench_sweep rows in one line each: python tools/sweep_brief.py <config> [...]  (EVREP_BIN_* switches apply)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import bench_sweep

for d in bench_sweep.sweep([a for a in sys.argv[1:] if a in bench_sweep.CONFIGS]):
    print("%-9s %-22s pass %d  bin %.4f  build %.4f  sum %.4f ms  (%.2f of 8 TB/s)" % (
        d["config"], d["builder"], d["binning_pass"], d["bin_ms"], d["build_ms"], d["bin_ms"] + d["build_ms"], d["build_frac_of_8TBps"]))
