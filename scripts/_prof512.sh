#!/bin/bash
# kernel trace of the 512-sample bf16-storage bench (3 steps), per-kernel summary under gpurun_out/
set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_b512s -o b -- python3 $ROOT/bench.py --storage bf16 --batch 512 --steps 3 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/prof_b512s.log 2>&1
python3 $ROOT/scripts/profile_summary.py $ROOT/gpurun_out/prof_b512s/b_kernel_trace.csv 3 > $ROOT/gpurun_out/r03_bench_B512_bf16_storage_per_kernel.txt
head -12 $ROOT/gpurun_out/r03_bench_B512_bf16_storage_per_kernel.txt
