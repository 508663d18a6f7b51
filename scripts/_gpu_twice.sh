#!/bin/bash
# The full GPU suite twice in a row (separate processes), verbose logs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
  timeout -k 10 500 python -u -m pytest tests -x -v --timeout 180 --timeout-method thread -m gpu > gpurun_out/rep_$i.log 2>&1
  rc=$?
  tail -1 gpurun_out/rep_$i.log
  if [ $rc -ne 0 ]; then echo "run $i failed rc=$rc"; grep -n "FAILED\|Error\|Fatal" gpurun_out/rep_$i.log | head; tail -30 gpurun_out/rep_$i.log; exit $rc; fi
done
