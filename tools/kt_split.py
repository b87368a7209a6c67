#!/usr/bin/env python3
"""Average duration per (kernel, grid) from a rocprofv3 kernel trace csv:  python tools/kt_split.py <dir> [name filter]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = [r for r in csv.DictReader(open(f)) if flt in r["Kernel_Name"]]
agg = collections.OrderedDict()
for r in rows:
    grid = r.get("Grid_Size") or "%sx%sx%s" % (r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
    k = (r["Kernel_Name"][:110], grid)
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print("%-112s grid %-14s calls %4d avg %8.1f us" % (k[0], k[1], len(v), sum(v) / len(v)))
