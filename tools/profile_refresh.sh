#!/bin/bash
# The sweep / clustered-counter / timeline part of tools/profile_round.sh alone (after a builder change that leaves the headline
# launch untouched): gpurun --timeout 1500 -- 'timeout 1400 bash tools/profile_refresh.sh', then bash tools/collect_profiles.sh rNN
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 900 python tools/bench_sweep.py > $O/sweep.jsonl 2> $O/sweep.err
timeout 900 python tools/bench_sweep.py gen1@circle gen1@edges c2@circle c2@edges c3@circle c3@edges > $O/sweep_clustered.jsonl 2>> $O/sweep.err
EVREP_X_VOXEL_ORDERED=1 EVREP_X_TORE_ORDERED=1 EVREP_X_POLSTATS_ORDERED=1 EVREP_X_ESTACK_ORDERED=1 EVREP_X_MDES_ORDERED=1 EVREP_X_TS_ORDERED=1 timeout 900 python tools/bench_sweep.py gen1 c2-dense gen1@circle gen1@edges c3@circle > $O/sweep_ordered_builders.jsonl 2>> $O/sweep.err
cd /tmp && export TMPDIR=/tmp
cd $R
bash tools/pmc_clustered.sh gen1 gen1@circle gen1@edges c3@circle c2-dense > $O/pmc_clustered.log 2>&1
bash tools/experiments/wave_lifetimes.sh > $O/wave_lifetimes.txt 2>&1
for cfg in "304,240,50000,32 uniform" "304,240,50000,32 circle" "304,240,50000,32 edges" "1280,720,200000,8 circle"; do set -- $cfg; EVREP_LIB_PATH=tools/variants/libevrep_timing.so SHAPE=$1 DIST=$2 timeout 100 python tools/experiments/wave_timeline.py 2>&1 | grep -v amdgpu; done > $O/wave_timeline.txt
EVREP_LIB_PATH=tools/variants/libevrep_timing.so SHAPE=640,480,500000,8 NBUF=1 timeout 300 python tools/experiments/phase_times.py 0 > $O/phase_times_dense.txt 2>&1
EVREP_LIB_PATH=tools/variants/libevrep_timing.so SHAPE=304,240,50000,32 NBUF=1 timeout 300 python tools/experiments/phase_times.py 0 > $O/phase_times_gen1.txt 2>&1
