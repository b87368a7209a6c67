#!/bin/bash
# Kernel trace of any command: gpurun -- 'bash tools/kt_cmd2.sh <tag> <command ...>' -> gpurun_out/<tag>/kernel_stats.csv + a table
TAG=${1:-kt}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- "$@" > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print("%-100s calls %6s avg %9.2f us  min %9.2f  max %9.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
