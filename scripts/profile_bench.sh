#!/bin/bash
# Kernel trace of the default bench command plus the two memory-side counter passes (separate runs), summaries
# written under gpurun_out/.  Run from the repository root on the GPU box.
set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_bench -o b -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/gpurun_out/prof_bench.log 2>&1
python3 $ROOT/scripts/profile_summary.py $ROOT/gpurun_out/prof_bench/b_kernel_trace.csv 10 > $ROOT/gpurun_out/r03_bench_per_kernel.txt
echo "trace done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_fetch -o f -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --eager > $ROOT/gpurun_out/pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_write -o w -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --eager > $ROOT/gpurun_out/pmc_write.log 2>&1
echo "write done"
python3 $ROOT/scripts/pmc_summary.py $ROOT/gpurun_out/pmc_fetch/f_counter_collection.csv conv_ tail_ head_ > $ROOT/gpurun_out/r03_pmc_fetch.txt
python3 $ROOT/scripts/pmc_summary.py $ROOT/gpurun_out/pmc_write/w_counter_collection.csv conv_ tail_ head_ > $ROOT/gpurun_out/r03_pmc_write.txt
