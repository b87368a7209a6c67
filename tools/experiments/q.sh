#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
export SWEEP_SHAPES="c2-1M:640,480,1000000,4;wide:2048,256,400000,8;tall:256,1024,200000,8"
timeout 600 python tools/sweep_table.py c3-1M c2-1M wide tall b=event_stack_f32
FUZZ_BIG=1 timeout 300 python tools/fuzz_campaign.py 900000 120 2>&1 | tail -2
timeout 300 python tools/fuzz_campaign.py 910000 60 2>&1 | tail -2
