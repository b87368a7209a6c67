#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_key_sorted.py tests/test_gpu_fuzz.py tests/test_gpu_clustered.py tests/test_gpu_builders.py tests/test_gpu_reference_api.py -x -q 2>&1 | tail -5
timeout 600 python tools/sweep_table.py gen1 c2 c3 gen1@circle c2@circle c3@circle gen1@edges c3@edges c2-150k c2-250k c2-dense b=event_stack_f32
