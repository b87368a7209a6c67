for cfg in "768" "1024"; do
  set -- $cfg
  echo "== hotmin $1"
  EVREP_VS_HOTMIN=$1 python tools/bench_sweep.py gen1 gen1@circle gen1@edges c2 c2@circle c3@circle c3@edges c2-dense b=voxel5_f64 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  %-8s %-8s build %7.1f us' % (d['config'], d['distribution'], d['build_ms']*1e3))
"
done
