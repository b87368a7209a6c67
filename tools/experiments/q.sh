F="--steps 300 --warmup 30 --no-cpu-baseline --no-gwd --no-gw-extension --no-live-traffic --no-sweep --no-precompute"
for i in 1 2; do
python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage64 ', round(d['ms_per_step']*1e3,1), round(d['roofline']['avg_launch_ms']*1e3,1))"
EVREP_X_STAGE128=1 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage128', round(d['ms_per_step']*1e3,1), round(d['roofline']['avg_launch_ms']*1e3,1))"
done
EVREP_X_STAGE128=1 python tools/sweep_table.py c2 c2@circle c2@edges b=optimized_f64 b=event_stack_f32 b=time_surface_f64 b=voxel5_f64
