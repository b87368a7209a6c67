#!/bin/bash
timeout 400 python tools/fuzz_campaign.py 1000000 300 2>&1 | tail -2
FUZZ_BIG=1 timeout 400 python tools/fuzz_campaign.py 1100000 300 2>&1 | tail -2
