#!/bin/bash
# Full GPU suite (verbose log), then the default bench run (the line the driver records).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -v --timeout 180 --timeout-method thread -m gpu > gpurun_out/full_t.log 2>&1
rc=$?
tail -3 gpurun_out/full_t.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; grep -n "FAILED\|Error" gpurun_out/full_t.log | head -20; exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err || exit 1
cat gpurun_out/r03_bench_line.json
