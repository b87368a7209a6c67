echo "== stage96"; EVREP_X_STAGE96=1 timeout 300 python tools/sweep_table.py gen1
echo "== default"; timeout 300 python tools/sweep_table.py gen1
EVREP_X_STAGE96=1 timeout 300 python -m pytest tests/test_gpu_builders.py tests/test_gpu_key_sorted.py -x -q 2>&1 | tail -2
