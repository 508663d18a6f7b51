#!/bin/bash
# the training step at every stage of the progression with the reference's minibatch per stage (one bench line each)
set -e
mkdir -p gpurun_out
: > gpurun_out/r03_depth_sweep.txt
for cfg in "0 256" "1 256" "2 128" "3 128" "4 48" "5 32" "6 32" "7 32"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --depth $1 --batch $2 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/_sweep.json 2> gpurun_out/_sweep.err || { tail -5 gpurun_out/_sweep.err; exit 1; }
  tail -1 gpurun_out/_sweep.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth $1 batch $2: %.1f steps/s  %.3f ms/step  whole-step frac %s' % (d['value'], d['ms_per_step'], d['roofline'].get('whole_step',{}).get('frac')))" | tee -a gpurun_out/r03_depth_sweep.txt
done
