cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "c2 circle optimized_f64" "gen1 circle optimized_f64" "c2 circle event_stack_f32"; do
  set -- $cfg
  rm -rf /tmp/hp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp -o hp -- python $R/tools/experiments/hot_prof.py $1 $2 $3 > /dev/null 2>&1
  echo "== $cfg"; python - <<PY
import csv,glob
f=glob.glob('/tmp/hp/**/hp_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['TotalDurationNs'])>2e5: print(r['Name'][:70], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1))
PY
done
rm -rf /tmp/hp; rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/hp -o hp -- python $R/tools/experiments/hot_prof.py c2 circle optimized_f64 4 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/hp/**/hp_counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:60]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k,v in acc.items():
    if 'k_mdes' in k: print(k, n[k], {c: round(x/max(n[k],1)) for c,x in v.items()})
PY
