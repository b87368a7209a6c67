#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every kernel of a BUILT library or object (no recompilation): the gfx950 code object is
cut out of the fat binary (__CLANG_OFFLOAD_BUNDLE__ / ELF magic scan) and its kernel metadata read with llvm-readelf.
   python tools/so_resources.py [path, default the package's libevrep.so] [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, "event_representation_study_amd", "libevrep.so")
flt = [a for a in sys.argv[1:] if not os.path.exists(a)]
data = open(path, "rb").read()
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
rows = []
pos = 0
while True:
    i = data.find(b"\x7fELF", pos)
    if i < 0:
        break
    pos = i + 4
    if i == 0 or data[i + 18:i + 20] != b"\xe0\x00":      # e_machine EM_AMDGPU = 224
        continue
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[i:])
        tmp = f.name
    out = subprocess.run([READELF, "--notes", tmp], capture_output=True, text=True).stdout
    os.remove(tmp)
    cur = {}
    for line in out.splitlines():
        m = re.match(r"\s+[-\s]\s*\.(\w+):\s+(.*)", line)
        if not m:
            m = re.match(r"\s+\.(\w+):\s+(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "name" and line.lstrip().startswith("- ") is False and v.startswith("_Z"):
            cur["name"] = v
        if k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count"):
            cur[k] = int(v)
        if k == "symbol":
            cur["symbol"] = v.strip("'")
        if k == "wavefront_size":        # the last key of a kernel's (alphabetical) metadata map
            rows.append(cur)
            cur = {}
names = subprocess.run(["c++filt"], input="\n".join(r.get("symbol", "?").replace(".kd", "") for r in rows), capture_output=True, text=True).stdout.splitlines()
print("%-86s %5s %5s %7s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("evrep::", "").replace("void ", "")
    if flt and not all(f in n for f in flt):
        continue
    print("%-86s %5d %5d %7d %6d" % (n[:86], r.get("vgpr_count", -1), r.get("sgpr_count", -1),
                                     r.get("private_segment_fixed_size", -1), r.get("group_segment_fixed_size", -1)))
