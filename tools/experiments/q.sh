timeout 600 python -m pytest tests/test_gpu_key_sorted.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py tests/test_gpu_builders.py tests/test_gpu_clustered.py -x -q 2>&1 | tail -3
timeout 300 python tools/sweep_table.py c2-dense c3-1M
FUZZ_BIG=1 timeout 200 python tools/fuzz_campaign.py 910000 150 2>&1 | tail -1
