#!/bin/bash
# same-session comparison of the committed tree (worktree _base, built beforehand) with the working tree, two rounds
set -e
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
for i in 1 2; do
for t in _base .; do
cd $ROOT/$t
timeout -k 10 200 python bench.py --no-cpu-baseline $AB_ARGS > $ROOT/gpurun_out/ab_tree.log 2>&1 || { tail -20 $ROOT/gpurun_out/ab_tree.log; exit 1; }
tail -1 $ROOT/gpurun_out/ab_tree.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['library'], d['value'], d['ms_per_step'])"
done
done
