#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_key_sorted.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py tests/test_gpu_reference_api.py -x -q 2>&1 | tail -3
timeout 600 python tools/sweep_table.py c3-1M c3 b=optimized_f64 b=event_stack_f32
timeout 300 python tools/per_sample_latency.py 2>&1 | tail -8
