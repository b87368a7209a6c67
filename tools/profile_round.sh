#!/bin/bash
# One measurement session for profiles/rNN (run on the GPU box through gpurun, from the repository root):
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh'
# bench (f64 with live PMC traffic, f32), rocprofv3 kernel trace + stats of the bench step and of the GWD / GW legs,
# the PMC passes (separate, as MI355X_MICROARCH.md asks), the builder sweep, per-sample latency, the GWD matrix, the
# EST bench and the precompute pipeline -> gpurun_out/final/, from where the summaries are copied into
# profiles/rNN/ (see profiles/README.md).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gw-extension --no-live-traffic --no-sweep --no-precompute > $O/bench_driver_flags.json 2>> $O/bench.err
timeout 900 python bench.py --out-dtype f32 --no-cpu-baseline --no-gwd --no-gw-extension --no-live-traffic --no-sweep --no-precompute > $O/bench_f32.json 2>> $O/bench.err
timeout 900 python bench.py --pacing 0 --no-cpu-baseline --no-gwd --no-gw-extension --no-live-traffic --no-sweep --no-precompute > $O/bench_unpaced.json 2>> $O/bench.err
timeout 300 python tools/experiments/pacing.py ergo64 8 0 600 650 680 690 700 720 750 > $O/pacing.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --no-sweep --no-precompute --no-pipelined-value > $O/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o pmc_fetch -- python $R/tools/pmc_workload.py > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o pmc_write -- python $R/tools/pmc_workload.py > $O/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq -o pmc_sq -- python $R/tools/pmc_workload.py > $O/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/b_sq -o pmc_builders_sq -- python $R/tools/pmc_builders_workload.py > $O/b_sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/b_fetch -o pmc_builders_fetch -- python $R/tools/pmc_builders_workload.py > $O/b_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/b_write -o pmc_builders_write -- python $R/tools/pmc_builders_workload.py > $O/b_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/gwd_sq -o pmc_gwd -- python $R/tools/gwd_batch_prof.py 36 > $O/gwd_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gwd_kt -o gwd -- python $R/tools/gwd_batch_prof.py 144 > $O/gwd_kt.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/gw_sq -o pmc_gw -- python $R/tools/gw_bench.py --outer 1 --sinkhorn 5 > $O/gw_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gw_kt -o gw -- python $R/tools/gw_bench.py --outer 1 --sinkhorn 5 > $O/gw_kt.log 2>&1
cd $R
timeout 900 python tools/bench_sweep.py > $O/sweep.jsonl 2> $O/sweep.err
timeout 900 python tools/bench_sweep.py gen1@circle gen1@edges c2@circle c2@edges c3@circle c3@edges > $O/sweep_clustered.jsonl 2>> $O/sweep.err
# r06: the ordered builders beside the streams (A/B by plan flags), and counters on clustered streams
EVREP_X_VOXEL_ORDERED=1 EVREP_X_TORE_ORDERED=1 EVREP_X_POLSTATS_ORDERED=1 EVREP_X_ESTACK_ORDERED=1 EVREP_X_MDES_ORDERED=1 EVREP_X_TS_ORDERED=1 timeout 900 python tools/bench_sweep.py gen1 c2-dense gen1@circle gen1@edges c3@circle > $O/sweep_ordered_builders.jsonl 2>> $O/sweep.err
bash tools/pmc_clustered.sh gen1 gen1@circle gen1@edges c3@circle c2-dense > $O/pmc_clustered.log 2>&1
bash tools/experiments/wave_lifetimes.sh > $O/wave_lifetimes.txt 2>&1
for cfg in "304,240,50000,32 uniform" "304,240,50000,32 circle" "304,240,50000,32 edges" "1280,720,200000,8 circle"; do set -- $cfg; EVREP_LIB_PATH=tools/variants/libevrep_timing.so SHAPE=$1 DIST=$2 timeout 100 python tools/experiments/wave_timeline.py 2>&1 | grep -v amdgpu; done > $O/wave_timeline.txt
timeout 900 python tools/per_sample_latency.py > $O/per_sample.jsonl 2>&1
timeout 900 python tools/gwd_matrix.py > $O/gwd_matrix.log 2>&1
timeout 900 python tools/gwd_matrix.py --windows 24 > $O/gwd_matrix24.log 2>&1
timeout 900 python tools/est_bench.py > $O/est_bench.json 2>/dev/null
timeout 900 python tools/gw_bench.py > $O/gw_bench_f64.json 2>/dev/null
timeout 900 python tools/gw_bench.py --precision f32 > $O/gw_bench_f32.json 2>/dev/null
timeout 900 python tools/precompute_reps.py --samples 1024 2>/dev/null | grep "^{" > $O/precompute.json
timeout 120 tools/microbench/gwd_tile_phases > $O/gwd_tile_phases.txt 2>&1
EVREP_LIB_PATH=tools/variants/libevrep_timing.so SHAPE=640,480,500000,8 NBUF=1 timeout 300 python tools/experiments/phase_times.py 0 > $O/phase_times_dense.txt 2>&1
EVREP_LIB_PATH=tools/variants/libevrep_timing.so SHAPE=304,240,50000,32 NBUF=1 timeout 300 python tools/experiments/phase_times.py 0 > $O/phase_times_gen1.txt 2>&1
EVREP_LIB_PATH=tools/variants/libevrep_timing.so NBUF=2 timeout 300 python tools/experiments/phase_times.py 0 690 > $O/phase_times.txt 2>&1
timeout 200 python tools/experiments/ks_phases.py > $O/ks_phases.txt 2>&1
timeout 200 python tools/experiments/tmpfs_write_floor.py > $O/tmpfs_write_floor.txt 2>&1
timeout 200 python tools/experiments/hot_overflow.py 720 1280 200000 16 > $O/hot_overflow.txt 2>&1
find $O -name "*.csv" | head -40
tail -c 900 $O/bench.json
