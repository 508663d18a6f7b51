#!/bin/bash
# same-session comparison of two settings of one environment variable (AB_VAR, values AB_VALS), two rounds
set -e
for i in 1 2; do
for val in ${AB_VALS:-0 1}; do
env $AB_VAR=$val timeout -k 10 200 python bench.py --no-cpu-baseline $AB_ARGS > gpurun_out/ab_$val.log 2>&1 || { tail -20 gpurun_out/ab_$val.log; exit 1; }
tail -1 gpurun_out/ab_$val.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$AB_VAR=$val', d['value'], d['ms_per_step'])"
done
done
