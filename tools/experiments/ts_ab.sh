for m in 1000000000 0; do
  echo "== EVREP_TSS_MIN $m"
  EVREP_TSS_MIN=$m python tools/bench_sweep.py gen1 gen1@circle gen1@edges c2 c2@circle c2@edges c3 c3@circle c3@edges c2-dense b=time_surface_f64 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  %-8s %-8s %-14s build %7.1f us  frac %.3f' % (d['config'], d['distribution'], d['builder'], d['build_ms']*1e3, d['build_frac_of_8TBps']))
"
done
