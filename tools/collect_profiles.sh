#!/bin/bash
# Copy the outputs of one tools/profile_round.sh session (gpurun_out/final) into profiles/<round> and refresh traffic.json.
#   bash tools/collect_profiles.sh r02
set -e
R=${1:?round directory name, e.g. r02}; O=gpurun_out/final; P=profiles/$R; mkdir -p $P
cp $O/bench.json $P/bench.json; cp $O/bench_f32.json $P/bench_f32.json; cp $O/bench_unpaced.json $O/bench_driver_flags.json $P/
cp $O/pacing.txt $P/pacing_final_kernel.txt
cp $O/kt/bench_kernel_stats.csv $P/bench_kernel_stats.csv; cp $O/kt/bench_kernel_trace.csv $P/bench_kernel_trace.csv
grep "^{" $O/kt.log > $P/bench_under_rocprof.json
cp $O/fetch/pmc_fetch_counter_collection.csv $O/write/pmc_write_counter_collection.csv $O/sq/pmc_sq_counter_collection.csv $P/
cp $O/b_sq/pmc_builders_sq_counter_collection.csv $O/b_fetch/pmc_builders_fetch_counter_collection.csv $O/b_write/pmc_builders_write_counter_collection.csv $P/
cp $O/gwd_sq/pmc_gwd_counter_collection.csv $O/gw_sq/pmc_gw_counter_collection.csv $O/gw_kt/gw_kernel_stats.csv $O/gwd_kt/gwd_kernel_stats.csv $P/
cp $O/sweep.jsonl $P/sweep.jsonl; cp $O/sweep_clustered.jsonl $P/sweep_clustered.jsonl; grep -v amdgpu.ids $O/wave_lifetimes.txt > $P/wave_lifetimes.txt; cp $O/wave_timeline.txt $P/wave_timeline.txt; [ -f $O/wave_timeline_tore.txt ] && cat $O/wave_timeline_tore.txt >> $P/wave_timeline.txt; grep "^{" $O/per_sample.jsonl > $P/per_sample_latency.jsonl
grep "^{" $O/gwd_matrix.log | tail -1 > $P/gwd_matrix.json || true
grep "^{" $O/gwd_matrix24.log | tail -1 > $P/gwd_matrix_24windows.json || true
cp $O/est_bench.json $O/gw_bench_f64.json $O/gw_bench_f32.json $O/precompute.json $P/
python tools/parse_pmc.py $P/pmc_fetch_counter_collection.csv $P/pmc_write_counter_collection.csv profiles/traffic.json > /dev/null
cp profiles/traffic.json $P/traffic.json
grep -v amdgpu.ids $O/gwd_tile_phases.txt > $P/gwd_tile_phases.txt || true
for f in phase_times phase_times_dense phase_times_gen1; do grep -v amdgpu.ids $O/$f.txt > $P/$f.txt || true; done
for f in ks_phases tmpfs_write_floor hot_overflow; do [ -f $O/$f.txt ] && grep -v amdgpu.ids $O/$f.txt > $P/$f.txt || true; done
# r06: ordered-builder A/B sweep and the clustered PMC passes (tools/pmc_clustered.sh writes gpurun_out/pmc_clustered/<stream>/)
[ -f $O/sweep_ordered_builders.jsonl ] && cp $O/sweep_ordered_builders.jsonl $P/ || true
if [ -d gpurun_out/pmc_clustered ]; then
  for d in gpurun_out/pmc_clustered/*/; do n=$(basename $d); mkdir -p $P/pmc_clustered/$n; cp $d/sq.csv $d/fetch.csv $d/write.csv $d/summary.txt $P/pmc_clustered/$n/ 2>/dev/null || true; done
fi
