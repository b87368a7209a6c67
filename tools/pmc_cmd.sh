#!/bin/bash
# PMC counters for any command (own pass, kernel-trace only): gpurun -- 'bash tools/pmc_cmd.sh <tag> "<counters>" <command...>'
TAG=$1; shift; CTR=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $O/pmc -o p -- "$@" > $O/pmc.log 2>&1
f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
cp $f $O/counter_collection.csv
python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, "n=%d" % len(next(iter(v.values()))))
PY
