#!/usr/bin/env python3
"""Per-kernel VGPR / scratch / occupancy / LDS of libevrep as hipcc reports them (no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "event_representation_study_amd", "csrc")
# usage: kernel_resources.py [translation unit, default evrep_capi_mdes.hip] [extra hipcc flags]
unit = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".hip") else "evrep_capi_mdes.hip"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
       "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", os.path.join(CSRC, unit), "-o", "/tmp/_kr.o",
       "-Rpass-analysis=kernel-resource-usage"] + [a for a in sys.argv[1:] if not a.endswith(".hip")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("evrep::", "")[:60]}
        rows.append(cur)
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split()[0]] = int(m.group(2))
print("%-62s %5s %5s %7s %4s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for r in rows:
    print("%-62s %5d %5d %7d %4d %6d" % (r["name"], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1),
                                        r.get("Occupancy", -1), r.get("LDS", -1)))
