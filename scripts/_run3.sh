set -e
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_b512s -o b -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 512 --storage bf16 > $ROOT/gpurun_out/prof_b512s.log 2>&1
python3 $ROOT/scripts/profile_summary.py $ROOT/gpurun_out/prof_b512s/b_kernel_trace.csv 3 > $ROOT/gpurun_out/r02_b512_bf16s_per_kernel.txt
head -45 $ROOT/gpurun_out/r02_b512_bf16s_per_kernel.txt
