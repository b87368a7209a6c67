timeout 400 python -m pytest tests/test_gpu_clustered.py tests/test_gpu_key_sorted.py tests/test_gpu_fuzz.py tests/test_gpu_builders.py -x -q 2>&1 | tail -3
timeout 300 python tools/sweep_table.py gen1 c2 c2-150k c2-250k c3 c2-dense c3-1M gen1@circle c2@circle c3@circle b=optimized_f64 b=event_stack_f32
