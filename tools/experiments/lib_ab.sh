#!/bin/bash
# A/B of library VARIANTS (compile-time knobs) on sweep tags, alternating the libraries twice (boxes drift by ~10 % between runs):
#   bash tools/experiments/lib_ab.sh "gen1 gen1@circle" tools/variants/libevrep_a.so tools/variants/libevrep_b.so ...   (the in-tree library runs first)
TAGS=$1; shift
mkdir -p gpurun_out/lib_ab
for rnd in 1 2; do
  timeout 600 python tools/bench_sweep.py $TAGS 2>/dev/null | grep "^{" > gpurun_out/lib_ab/base_$rnd.jsonl
  for L in "$@"; do
    EVREP_LIB_PATH=$L timeout 600 python tools/bench_sweep.py $TAGS 2>/dev/null | grep "^{" > gpurun_out/lib_ab/$(basename $L .so)_$rnd.jsonl
  done
done
python - "$@" <<'PY'
import json, sys, os, glob
names = ["base"] + [os.path.basename(p)[:-3] for p in sys.argv[1:]]
rows = {}
for n in names:
    for f in sorted(glob.glob("gpurun_out/lib_ab/%s_*.jsonl" % n)):
        for line in open(f):
            d = json.loads(line)
            rows.setdefault((d["config"], d["distribution"], d["builder"]), {}).setdefault(n, []).append(d["build_ms"] * 1e3)
print("%-8s %-8s %-22s " % ("config", "dist", "builder") + " ".join("%16s" % n[-16:] for n in names))
for k, v in rows.items():
    print("%-8s %-8s %-22s " % k + " ".join("%16s" % ("/".join("%.1f" % x for x in v.get(n, []))) for n in names))
PY
