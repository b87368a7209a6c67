timeout 400 python -m pytest tests/test_gpu_clustered.py tests/test_gpu_key_sorted.py tests/test_gpu_fuzz.py tests/test_gpu_builders.py tests/test_gpu_properties.py -x -q 2>&1 | tail -3
timeout 300 python tools/sweep_table.py gen1@circle gen1@edges c2@circle c2@edges c3@circle b=optimized_f64 b=optimized_f32 b=event_stack_f32 b=time_surface_f64 b=voxel5_f64 b=tore_full_frame_f32
timeout 300 python tools/sweep_table.py gen1 c2 c3 b=optimized_f64 b=optimized_f32 b=event_stack_f32 b=time_surface_f64
