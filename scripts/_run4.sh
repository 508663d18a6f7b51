set -o pipefail
timeout -k 10 900 python -u -m pytest tests/test_ops_gpu.py tests/test_bf16_mode.py tests/test_bf16_storage.py tests/test_e2e_gpu.py -x -v --timeout 300 --timeout-method thread > gpurun_out/s2_t.log 2>&1
rc=$?
grep -n "PASSED\|FAILED\|ERROR" gpurun_out/s2_t.log | tail -2
grep -n "Fatal\|fault\|Abort\|Segmentation\|Timeout" gpurun_out/s2_t.log | head -20
[ $rc -eq 0 ] || { tail -40 gpurun_out/s2_t.log; exit $rc; }
bash scripts/_ab.sh
