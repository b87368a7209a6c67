T="c2 gen1@circle c2@circle c3@circle b=optimized_f64 b=event_stack_f32"
echo "== default"; timeout 300 python tools/sweep_table.py $T
for v in g1024 g512; do echo "== $v"; EVREP_LIB_PATH=tools/variants/$v.so timeout 300 python tools/sweep_table.py $T; done
F="--steps 300 --warmup 30 --no-cpu-baseline --no-gwd --no-gw-extension --no-live-traffic --no-sweep --no-precompute"
for v in "" tools/variants/g1024.so tools/variants/g512.so; do EVREP_LIB_PATH=$v python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step']*1e3,1), d['step_us'])"; done
