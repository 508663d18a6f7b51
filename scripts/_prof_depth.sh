#!/bin/bash
# kernel trace of one stage of the progression (DEPTH, BATCH), per-kernel summary under gpurun_out/
set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_depth -o b -- python3 $ROOT/bench.py --depth $DEPTH --batch $BATCH --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/gpurun_out/prof_depth.log 2>&1
python3 $ROOT/scripts/profile_summary.py $ROOT/gpurun_out/prof_depth/b_kernel_trace.csv 10 > $ROOT/gpurun_out/r03_depth${DEPTH}_per_kernel.txt
head -30 $ROOT/gpurun_out/r03_depth${DEPTH}_per_kernel.txt
