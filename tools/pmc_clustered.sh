#!/bin/bash
# PMC passes (SQ set, FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) over tools/pmc_builders_workload.py for
# CLUSTERED streams, one rocprofv3 run per (stream, counter set) so the kernels of different shapes stay apart:
#   gpurun --timeout 1500 -- 'bash tools/pmc_clustered.sh gen1@circle gen1@edges c3@circle'
# -> gpurun_out/pmc_clustered/<stream>/{sq,fetch,write}.csv + summary.txt (tools/pmc_clustered_summary.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_clustered; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
for tag in "$@"; do
  d=$O/${tag/@/_}; mkdir -p $d
  for set in ${PMC_SETS:-sq fetch write}; do
    case $set in sq) C="$SQ";; fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; esac
    timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d/$set -o p -- python $R/tools/pmc_builders_workload.py $tag $PMC_BUILDERS > $d/$set.log 2>&1
    f=$(find $d/$set -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp $f $d/$set.csv
    rm -rf $d/$set
  done
  python3 $R/tools/pmc_clustered_summary.py $d $tag | tee $d/summary.txt
done
