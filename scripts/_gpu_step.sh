#!/bin/bash
# One GPU session: the tests named in $TESTS (verbose log under gpurun_out/), then optional extras.
set -o pipefail
mkdir -p gpurun_out
echo "tests: $TESTS"
timeout -k 10 ${TEST_LIMIT:-600} python -u -m pytest $TESTS -x -v --timeout 180 --timeout-method thread -m gpu -s > gpurun_out/s_t.log 2>&1
rc=$?
tail -5 gpurun_out/s_t.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; grep -n "FAILED\|Error\|error" gpurun_out/s_t.log | head -20; exit $rc; fi
if [ -n "$TRACE" ]; then
  timeout -k 5 60 ./scripts/conv_trace 64 1920 > gpurun_out/conv_trace.txt 2>&1 || exit 1
  head -3 gpurun_out/conv_trace.txt
fi
