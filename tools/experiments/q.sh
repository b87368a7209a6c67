timeout 400 python -m pytest tests/test_gpu_clustered.py tests/test_gpu_key_sorted.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
T="gen1@circle gen1@edges c2@circle c3@circle b=optimized_f64 b=event_stack_f32 b=time_surface_f64"
echo "== default (hot stage 256)"; timeout 300 python tools/sweep_table.py $T
for v in s512 s768; do echo "== $v"; EVREP_LIB_PATH=tools/variants/$v.so timeout 300 python tools/sweep_table.py $T; done
EVREP_LIB_PATH=tools/variants/libevrep_timing.so SHAPE=304,240,50000,32 DIST=circle NBUF=1 TOP=4 timeout 200 python tools/experiments/phase_times.py -1 2>&1 | grep -v amdgpu | cut -c1-200
