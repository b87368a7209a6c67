#!/bin/bash
# Quick kernel-trace of the bench step: gpurun -- 'bash tools/kt.sh <tag> [bench args]'  -> gpurun_out/<tag>/kernel_stats.csv
TAG=${1:-kt}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gwd --no-gw-extension --no-live-traffic "$@" > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s calls %6s avg %9.2f us  min %9.2f  max %9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
grep "^{" $O/kt.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'builder', d['roofline']['avg_launch_ms'])"
