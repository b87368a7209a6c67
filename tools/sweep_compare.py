#!/usr/bin/env python3
"""Table of a bench_sweep .jsonl (bin / build us, fraction of 8 TB/s, bin + build over the uniform row of the same shape and
builder), optionally beside an earlier run:  python tools/sweep_compare.py new.jsonl [old.jsonl]"""
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
old = {}
if len(sys.argv) > 2:
    for l in open(sys.argv[2]):
        if l.startswith("{"):
            r = json.loads(l)
            old[(r["config"], r["distribution"], r["builder"])] = r["build_ms"]
uni = {(r["config"], r["builder"]): r["bin_ms"] + r["build_ms"] for r in rows if r["distribution"] == "uniform"}
for r in rows:
    k = (r["config"], r["builder"])
    tot = r["bin_ms"] + r["build_ms"]
    o = old.get((r["config"], r["distribution"], r["builder"]))
    print("%-8s %-8s %-24s pass%d bin %6.1f build %6.1f %s frac %.3f x_uniform %.2f" % (
        r["config"], r["distribution"], r["builder"], r["binning_pass"], r["bin_ms"] * 1e3, r["build_ms"] * 1e3,
        ("(was %6.1f)" % (o * 1e3)) if o else "", r["build_frac_of_8TBps"], tot / uni.get(k, tot)))
