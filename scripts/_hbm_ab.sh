#!/bin/bash
set -e
mkdir -p gpurun_out
out=gpurun_out/hbm_ab.txt
: > $out
timeout -k 10 200 python scripts/bench_hbm.py 512 64 32 2>&1 | grep -v amdgpu.ids | tee -a $out
