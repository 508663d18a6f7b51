#!/bin/bash
# Measurement pass of the build in the tree (run on the GPU box from the repository root): kernel trace and counter
# passes of the bench command, counter passes of the stand-alone launch, step parts, the default bench line (which
# then reads this session's summaries for its traffic and in-step figures), the side configurations, the per-kernel
# roofline table, the stage sweep and the evaluation-metrics workload.  Everything lands in gpurun_out/ as r03_*;
# scripts/collect_profiles.sh copies those into profiles/.
set -e
mkdir -p gpurun_out
bash scripts/profile_bench.sh
cp gpurun_out/r03_bench_per_kernel.txt profiles/r03_bench_per_kernel.txt
bash scripts/pmc_conv_ws.sh
timeout -k 10 300 python scripts/bench_step_parts.py > gpurun_out/r03_step_parts.txt 2>&1 || tail -5 gpurun_out/r03_step_parts.txt
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
timeout -k 10 300 python bench.py --alpha 0.5 --no-cpu-baseline > gpurun_out/r03_bench_alpha05.json 2>/dev/null
timeout -k 10 300 python bench.py --depth 0 --no-cpu-baseline > gpurun_out/r03_bench_depth0_B32.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 512 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_B512_f32.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 512 --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 > gpurun_out/r03_bench_B512_bf16.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 512 --steps 10 --warmup 3 --no-cpu-baseline --storage bf16 > gpurun_out/r03_bench_B512_bf16_storage.json 2>/dev/null
python - <<'PY'
import json
for n in ("r03_bench_line", "r03_bench_alpha05", "r03_bench_depth0_B32", "r03_bench_B512_f32", "r03_bench_B512_bf16", "r03_bench_B512_bf16_storage"):
    d = json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(n, d["value"], d["ms_per_step"], r.get("frac"), r.get("in_step", {}).get("chain", {}).get("frac"),
          r.get("in_step", {}).get("frac"), r.get("whole_step", {}).get("frac"), r.get("traffic_over_algorithmic"))
PY
timeout -k 10 400 python scripts/kernel_roofline.py 32 64 512 > gpurun_out/r03_kernel_roofline_B32_B64_B512.txt 2>&1
timeout -k 10 400 python scripts/bench_hbm.py 32 64 512 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_hbm_kernels_rotating_buffers.txt
timeout -k 10 300 python scripts/bench_hbm.py 512 --bf16 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_hbm_kernels_rotating_buffers_bf16.txt
timeout -k 10 120 ./scripts/pn_probe 512 4 > gpurun_out/r03_probe_pixelnorm_B512.txt 2>&1
( timeout -k 10 120 ./scripts/rs_probe 512 1280 1920 && timeout -k 10 120 ./scripts/rs_probe 512 1920 1280 ) > gpurun_out/r03_probe_resampler_B512.txt 2>&1
bash scripts/_pmc_hbm.sh > /dev/null
bash scripts/_depth_sweep.sh
timeout -k 10 500 python scripts/bench_metrics.py > gpurun_out/r03_metrics_bench.json 2> gpurun_out/r03_metrics_bench.err || tail -3 gpurun_out/r03_metrics_bench.err
