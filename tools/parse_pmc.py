#!/usr/bin/env python3
"""Fold the rocprofv3 counter CSVs of tools/pmc_workload.py into profiles/traffic.json.

Corrections, as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes: FETCH_SIZE and
WRITE_SIZE are in KiB-like units derived from 64-byte request tallies and, on gfx950, FETCH_SIZE
counts a wide coalesced stream at half its bytes -- so both are CALIBRATED here on kernels of known
byte count (a 1 GiB fill and a 1 GiB copy issued by the same process) before being applied.
"""
import csv
import json
import os
import sys

CAL_BYTES = float(1 << 30)


def per_kernel(path, counter):
    acc = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return acc


def total(acc, needle):
    return sum(sum(v) for k, v in acc.items() if needle.lower() in k.lower())


def main(fetch_csv, write_csv, out_json):
    fetch = per_kernel(fetch_csv, "FETCH_SIZE")
    write = per_kernel(write_csv, "WRITE_SIZE")
    names = sorted(set(fetch) | set(write))
    # WRITE_SIZE: bytes per unit from the fill kernels (each writes exactly 1 GiB)
    fills = [v for k, v in write.items() if "fill" in k.lower()]
    n_fill = sum(len(v) for v in fills)
    w_unit = CAL_BYTES * n_fill / sum(sum(v) for v in fills)
    # FETCH_SIZE: the device-to-device copies read exactly what they write, so bytes per FETCH unit =
    # (WRITE units of all copy dispatches * w_unit) / (FETCH units of all copy dispatches)
    f_unit = total(write, "copyBuffer") * w_unit / total(fetch, "copyBuffer")
    cal = {"fill_launches": n_fill, "bytes_per_WRITE_SIZE_unit": w_unit, "bytes_per_FETCH_SIZE_unit": f_unit,
           "note": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM): "
                   "expect ~2048 bytes per unit for wide streaming reads, 1024 for WRITE_SIZE"}
    res = {"calibration": cal, "kernels": {}}
    for n in names:
        fv = sum(fetch.get(n, [0])) / max(1, len(fetch.get(n, [])))
        wv = sum(write.get(n, [0])) / max(1, len(write.get(n, [])))
        res["kernels"][n[:120]] = {
            "FETCH_SIZE_raw": fv, "WRITE_SIZE_raw": wv,
            "read_bytes": fv * f_unit if f_unit else None, "write_bytes": wv * w_unit if w_unit else None}
    for key, needle in (("k_mdes_f64", "k_mdes<double"), ("k_mdes_f32", "k_mdes<float")):
        for n, v in res["kernels"].items():
            # the builder's MAIN launch (r04: every builder launch is followed by a hot launch of the same kernel template,
            # last template argument `true`, which is empty on these uniform windows)
            if needle in n and ", true>" not in n and v["read_bytes"] is not None and v["write_bytes"] is not None:
                res[key] = {"batch": 32, "events": 50000, "hbm_bytes_per_launch": v["read_bytes"] + v["write_bytes"],
                            "read_bytes": v["read_bytes"], "write_bytes": v["write_bytes"],
                            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), calibrated on a "
                                      "1 GiB fill and copy; see profiles/README.md"}
    with open(out_json, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in res if k.startswith("k_mdes")}, indent=1))
    print(json.dumps(cal, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
