cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "c2-dense uniform optimized_f64" "c3-1M uniform optimized_f64"; do
  set -- $cfg
  rm -rf /tmp/hp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp -o hp -- python $R/tools/experiments/hot_prof.py $1 $2 $3 > /dev/null 2>&1
  echo "== $cfg"; python - <<PY
import csv,glob
f=glob.glob('/tmp/hp/**/hp_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['TotalDurationNs'])>1e5: print(r['Name'][:75], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1))
PY
done
