#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/fuzz_campaign.py 700000 120 2>&1 | tail -2
FUZZ_BIG=1 timeout 300 python tools/fuzz_campaign.py 800000 150 2>&1 | tail -2
