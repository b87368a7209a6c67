#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_clustered.py -x -q -k "order_free" 2>&1 | tail -15
