#!/bin/bash
# Measurement pass of the headline configuration: trace + counter passes, step parts, then the default bench line
# (which reads the fresh per-kernel summary for its in-step figure).
set -e
mkdir -p gpurun_out
bash scripts/profile_bench.sh
cp gpurun_out/r03_bench_per_kernel.txt profiles/r03_bench_per_kernel.txt
timeout -k 10 300 python scripts/bench_step_parts.py > gpurun_out/r03_step_parts.txt 2>&1 || tail -5 gpurun_out/r03_step_parts.txt
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
timeout -k 10 300 python bench.py --alpha 0.5 --no-cpu-baseline > gpurun_out/r03_bench_alpha05.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 512 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_B512_f32.json 2>/dev/null
python - <<'PY'
import json
for n in ("r03_bench_line", "r03_bench_alpha05", "r03_bench_B512_f32"):
    d = json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"].get("frac"), d["roofline"].get("in_step", {}).get("frac"), d["roofline"].get("whole_step", {}).get("frac"))
PY
