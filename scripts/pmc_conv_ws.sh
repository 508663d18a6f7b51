#!/bin/bash
# Memory-side counter passes of the stand-alone 64 -> 64, k = 3, T = 1920, 32-sample launch of the weight-stationary
# kernel (one run per counter), written as profiles/r03_pmc_conv_ws_64x64_T1920_B32.csv for bench.py's roofline.traffic.
set -e
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_ws_fetch -o f -- python3 $ROOT/scripts/bench_one.py 32 64 64 1920 3 > $ROOT/gpurun_out/pmc_ws_fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_ws_write -o w -- python3 $ROOT/scripts/bench_one.py 32 64 64 1920 3 > $ROOT/gpurun_out/pmc_ws_write.log 2>&1
echo "write pass done"
F=$(find $ROOT/gpurun_out/pmc_ws_fetch -name '*counter_collection.csv' | head -1)
W=$(find $ROOT/gpurun_out/pmc_ws_write -name '*counter_collection.csv' | head -1)
python3 $ROOT/scripts/pmc_conv_ws.py $F $W > $ROOT/gpurun_out/r03_pmc_conv_ws_64x64_T1920_B32.csv
cp $ROOT/gpurun_out/r03_pmc_conv_ws_64x64_T1920_B32.csv $ROOT/profiles/r03_pmc_conv_ws_64x64_T1920_B32.csv
grep -c conv_ws $ROOT/profiles/r03_pmc_conv_ws_64x64_T1920_B32.csv
