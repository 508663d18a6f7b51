set -o pipefail
timeout -k 10 900 python -u -m pytest tests/test_variants.py -m gpu -x -v -s --timeout 300 --timeout-method thread > gpurun_out/s2_t.log 2>&1
rc=$?
grep -n "PASSED\|FAILED\|ERROR\|worst" gpurun_out/s2_t.log | tail -30
[ $rc -eq 0 ] || { tail -50 gpurun_out/s2_t.log; exit $rc; }
