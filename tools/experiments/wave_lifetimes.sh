#!/bin/bash
# Per-wave lifetimes of the ERGO-12 float64 builder (mean / p95 / p99 / max, microseconds) on uniform, moving-circle and
# edge-cluster windows, from the EVREP_TIMING build (tools/variants/libevrep_timing.so: hipcc ... -DEVREP_TIMING).
L=tools/variants/libevrep_timing.so
for cfg in "640,480,50000,32 uniform" "640,480,50000,32 circle" "640,480,50000,32 edges" "304,240,50000,32 uniform" "304,240,50000,32 circle" "304,240,50000,32 edges"; do
  set -- $cfg
  EVREP_LIB_PATH=$L SHAPE=$1 DIST=$2 NBUF=1 timeout 200 python tools/experiments/phase_times.py -1 2>&1 | grep -v amdgpu
done
