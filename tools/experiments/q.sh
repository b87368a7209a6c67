timeout 400 python -m pytest tests/test_gpu_clustered.py tests/test_gpu_key_sorted.py tests/test_gpu_fuzz.py tests/test_gpu_builders.py tests/test_gpu_reference_api.py -x -q 2>&1 | tail -3
echo "== merged tails"; timeout 300 python tools/sweep_table.py gen1 gen1@circle gen1@edges
echo "== no merge"; EVREP_NO_TAIL_MERGE=1 timeout 300 python tools/sweep_table.py gen1
