#!/usr/bin/env python3
"""Print bench_sweep JSON-lines files as a table; with two files, the second beside the first.
   python tools/sweep_show.py new.jsonl [old.jsonl]"""
import json
import sys


def load(p):
    rows = {}
    for line in open(p):
        if line.startswith("{"):
            d = json.loads(line)
            rows[(d["config"], d["distribution"], d["builder"])] = d
    return rows


new = load(sys.argv[1])
old = load(sys.argv[2]) if len(sys.argv) > 2 else {}
uni = {(k[0], k[2]): v for k, v in new.items() if k[1] == "uniform"}
for k, d in new.items():
    line = "%-8s %-8s %-22s pass %d bin %6.1f build %7.1f us frac %.3f" % (
        k[0], k[1], k[2], d["binning_pass"], d["bin_ms"] * 1e3, d["build_ms"] * 1e3, d["build_frac_of_8TBps"])
    u = uni.get((k[0], k[2]))
    if u is not None and k[1] != "uniform":
        line += "  x%.2f uniform" % ((d["bin_ms"] + d["build_ms"]) / (u["bin_ms"] + u["build_ms"]))
    o = old.get(k)
    if o is not None:
        line += "   | old bin %6.1f build %7.1f (%+.0f%%)" % (o["bin_ms"] * 1e3, o["build_ms"] * 1e3,
                                                            100 * (d["build_ms"] / o["build_ms"] - 1))
    print(line)
