#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (one run each) over a probe binary: bash scripts/_pmc_probe.sh <tag> <binary> [args]
set -e
ROOT=$(pwd)
tag=$1; shift
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_${tag}_fetch -o f -- $ROOT/"$@" > $ROOT/gpurun_out/pmc_${tag}_fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_${tag}_write -o w -- $ROOT/"$@" > $ROOT/gpurun_out/pmc_${tag}_write.log 2>&1
echo "write pass done"
F=$(find $ROOT/gpurun_out/pmc_${tag}_fetch -name '*counter_collection.csv' | head -1)
W=$(find $ROOT/gpurun_out/pmc_${tag}_write -name '*counter_collection.csv' | head -1)
python3 - "$F" "$W" <<'PY' | tee $ROOT/gpurun_out/pmc_${tag}.txt
import csv, sys, collections
acc = collections.OrderedDict()
for tagc, path in (("fetch", sys.argv[1]), ("write", sys.argv[2])):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][-70:]
        acc.setdefault(k, {}).setdefault(tagc, []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    f = v.get("fetch", [0]); w = v.get("write", [0])
    f = f[len(f) // 3:]; w = w[len(w) // 3:]
    fm = 2 * sum(f) / len(f) * 1024 / 1e6; wm = sum(w) / len(w) * 1024 / 1e6
    print(f"{k:72s} n={len(f):3d} read {fm:8.1f} MB  written {wm:8.1f} MB  total {fm + wm:8.1f} MB")
PY
