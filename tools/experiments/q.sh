O=gpurun_out/final; mkdir -p $O
bash tools/experiments/wave_lifetimes.sh > $O/wave_lifetimes.txt 2>&1
EVREP_LIB_PATH=tools/variants/libevrep_timing.so SHAPE=640,480,500000,8 NBUF=1 timeout 300 python tools/experiments/phase_times.py 0 > $O/phase_times_dense.txt 2>&1
EVREP_LIB_PATH=tools/variants/libevrep_timing.so SHAPE=304,240,50000,32 NBUF=1 timeout 300 python tools/experiments/phase_times.py 0 > $O/phase_times_gen1.txt 2>&1
EVREP_LIB_PATH=tools/variants/libevrep_timing.so NBUF=2 timeout 300 python tools/experiments/phase_times.py 0 690 > $O/phase_times.txt 2>&1
cat $O/wave_lifetimes.txt | cut -c1-150
