#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_key_sorted.py tests/test_gpu_fuzz.py tests/test_gpu_clustered.py tests/test_gpu_builders.py -x -q 2>&1 | tail -4
for i in 1 2; do
echo "--- prev"; EVREP_LIB_PATH=tools/variants/libevrep_prev.so timeout 600 python tools/sweep_table.py gen1@circle gen1@edges c2@circle c2@edges c3@circle b=optimized_f64 b=voxel5_f64 b=tore_full_frame_f32
echo "--- new"; timeout 600 python tools/sweep_table.py gen1@circle gen1@edges c2@circle c2@edges c3@circle b=optimized_f64 b=voxel5_f64 b=tore_full_frame_f32
done
