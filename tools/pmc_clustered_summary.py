#!/usr/bin/env python3
"""Fold one stream's PMC passes (tools/pmc_clustered.sh: sq.csv, fetch.csv, write.csv) into a table per evrep kernel:
launch time, waves, instructions per wave, where the wave cycles go (SQ_WAIT_ANY = parked on s_waitcnt / barrier,
SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_VALU = issuing VALU; quad-cycles, MI355X_MICROARCH.md SQ table), how busy the
chip's 1024 SIMDs are with VALU over the launch, and HBM traffic (FETCH_SIZE / WRITE_SIZE calibrated on the 1 GiB fill and
copy of the same process, as tools/parse_pmc.py does)."""
import collections
import csv
import os
import sys

CLK_GHZ, SIMDS = 2.4, 1024


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    if not os.path.exists(path):
        return acc, dur
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (r["Dispatch_Id"],)
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc, dur


def mean(v):
    return sum(v) / max(1, len(v))


def short(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    return k[:k.find("(")] if "(" in k else k


def main(d, tag):
    sq, dur = load(os.path.join(d, "sq.csv"))
    fe, _ = load(os.path.join(d, "fetch.csv"))
    wr, _ = load(os.path.join(d, "write.csv"))
    fills = [v["WRITE_SIZE"] for k, v in wr.items() if "fill" in k.lower()]
    w_unit = (float(1 << 30) * sum(len(v) for v in fills) / sum(sum(v) for v in fills)) if fills else 1024.0
    cw = sum(sum(v["WRITE_SIZE"]) for k, v in wr.items() if "copybuffer" in k.lower())
    cf = sum(sum(v["FETCH_SIZE"]) for k, v in fe.items() if "copybuffer" in k.lower())
    f_unit = cw * w_unit / cf if cf else 2048.0
    print("# %s   bytes per WRITE_SIZE unit %.0f, per FETCH_SIZE unit %.0f" % (tag, w_unit, f_unit))
    print("%-58s %5s %8s %8s %6s %6s %5s %6s %6s %6s %6s %8s %8s" % (
        "kernel", "n", "us", "waves", "valu/w", "salu/w", "lds/w", "wait", "stall", "valu", "simd", "readMB", "writeMB"))
    for k in sorted(sq, key=lambda k: -mean(dur[k])):
        if "k_" not in k or "at::" in k:
            continue
        c = {n: mean(v) for n, v in sq[k].items()}
        waves = max(1.0, c.get("SQ_WAVES", 1.0))
        wc = max(1.0, c.get("SQ_WAVE_CYCLES", 1.0))
        us = mean(dur[k])
        simd = c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (us * 1e3 * CLK_GHZ * SIMDS)
        rd = mean(fe[k]["FETCH_SIZE"]) * f_unit / 1e6 if k in fe else float("nan")
        wt = mean(wr[k]["WRITE_SIZE"]) * w_unit / 1e6 if k in wr else float("nan")
        print("%-58s %5d %8.1f %8d %6.0f %6.0f %5.0f %6.2f %6.2f %6.2f %6.2f %8.1f %8.1f" % (
            short(k)[:58], len(dur[k]), us, waves, c.get("SQ_INSTS_VALU", 0) / waves, c.get("SQ_INSTS_SALU", 0) / waves,
            c.get("SQ_INSTS_LDS", 0) / waves, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
            c.get("SQ_ACTIVE_INST_VALU", 0) / wc, simd, rd, wt))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
