#!/bin/bash
for i in 1 2; do
echo prev; EVREP_LIB_PATH=tools/variants/libevrep_prev.so timeout 300 python tools/experiments/pacing.py ts64 3 0 2>&1 | grep -v "amdgpu\|#"
echo new; timeout 300 python tools/experiments/pacing.py ts64 3 0 2>&1 | grep -v "amdgpu\|#"
echo prev; EVREP_LIB_PATH=tools/variants/libevrep_prev.so timeout 300 python tools/experiments/pacing.py ts64_1mpx 3 0 2>&1 | grep -v "amdgpu\|#"
echo new; timeout 300 python tools/experiments/pacing.py ts64_1mpx 3 0 2>&1 | grep -v "amdgpu\|#"
done
