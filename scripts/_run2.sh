set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q -s --timeout 300 --timeout-method thread > gpurun_out/s2_gpu_tests.log 2>&1 || { tail -60 gpurun_out/s2_gpu_tests.log; exit 1; }
grep -E "bf16 storage|passed|failed" gpurun_out/s2_gpu_tests.log | cut -c1-600
for mode in "--dtype f32" "--dtype bf16" "--storage bf16"; do
timeout -k 10 300 python bench.py --no-cpu-baseline --batch 512 --steps 8 --warmup 2 $mode > gpurun_out/s2_b512.log 2>&1 || { tail -30 gpurun_out/s2_b512.log; exit 1; }
tail -1 gpurun_out/s2_b512.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['bound'])"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --storage bf16 > gpurun_out/s2_b32_bf16s.log 2>&1 && tail -1 gpurun_out/s2_b32_bf16s.log | cut -c1-300
