#!/usr/bin/env python3
"""bench_sweep rows as one line each:  python tools/sweep_table.py <tags / b=builder ...>"""
import json
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
p = subprocess.run([sys.executable, os.path.join(here, "bench_sweep.py")] + sys.argv[1:], capture_output=True, text=True)
for line in p.stdout.splitlines():
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("%-8s %-8s %-22s pass %d bin %7.1f us  build %8.1f us  frac %.3f" % (
        d["config"], d["distribution"], d["builder"], d["binning_pass"], d["bin_ms"] * 1e3, d["build_ms"] * 1e3,
        d["build_frac_of_8TBps"]))
if p.returncode:
    sys.stderr.write(p.stderr[-2000:])
