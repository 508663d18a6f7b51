#!/bin/bash
# After scripts/_measure.sh ran on the GPU box: copy the measured summaries from gpurun_out/ into profiles/ (tracked).
set -e
cd "$(dirname "$0")/.."
for f in r03_bench_per_kernel.txt r03_pmc_conv_ws_64x64_T1920_B32.csv r03_step_parts.txt r03_bench_line.json \
         r03_bench_alpha05.json r03_bench_depth0_B32.json r03_bench_B512_f32.json r03_bench_B512_bf16.json \
         r03_bench_B512_bf16_storage.json r03_kernel_roofline_B32_B64_B512.txt r03_depth_sweep.txt r03_metrics_bench.json \
         r03_hbm_kernels_rotating_buffers.txt r03_probe_pixelnorm_B512.txt r03_probe_resampler_B512.txt \
         r03_pmc_hbm_kernels_B512.txt r03_hbm_kernels_rotating_buffers_bf16.txt; do
  [ -s gpurun_out/$f ] && cp gpurun_out/$f profiles/$f
done
[ -s gpurun_out/r03_pmc_fetch.txt ] && cp gpurun_out/r03_pmc_fetch.txt profiles/r03_pmc_fetch_bench_eager.txt
[ -s gpurun_out/r03_pmc_write.txt ] && cp gpurun_out/r03_pmc_write.txt profiles/r03_pmc_write_bench_eager.txt
[ -s gpurun_out/prof_bench/b_kernel_stats.csv ] && cp gpurun_out/prof_bench/b_kernel_stats.csv profiles/r03_bench_kernel_stats.csv
ls profiles/r03_*
