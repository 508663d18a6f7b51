set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/s2_gpu_tests.log 2>&1 || { tail -30 gpurun_out/s2_gpu_tests.log; exit 1; }
tail -3 gpurun_out/s2_gpu_tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/s2_bench_a.log 2>&1 && tail -1 gpurun_out/s2_bench_a.log | cut -c1-200
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/s2_bench_b.log 2>&1 && tail -1 gpurun_out/s2_bench_b.log | cut -c1-200
HIP_FORCE_DEV_KERNARG=0 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/s2_bench_c.log 2>&1 && tail -1 gpurun_out/s2_bench_c.log | cut -c1-200
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/s2_bench_d.log 2>&1 && tail -1 gpurun_out/s2_bench_d.log | cut -c1-200
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/s2_bench_e.log 2>&1 && tail -1 gpurun_out/s2_bench_e.log | cut -c1-200
