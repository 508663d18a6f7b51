#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (one run each) over the bandwidth-bound kernels at 512 samples (scripts/bench_hbm.py):
# bytes moved per launch against the algorithmic bytes, per kernel name -> gpurun_out/r03_pmc_hbm_kernels_B512.txt
set -e
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_hbm_fetch -o f -- python3 $ROOT/scripts/bench_hbm.py 512 > $ROOT/gpurun_out/pmc_hbm_fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_hbm_write -o w -- python3 $ROOT/scripts/bench_hbm.py 512 > $ROOT/gpurun_out/pmc_hbm_write.log 2>&1
echo "write pass done"
F=$(find $ROOT/gpurun_out/pmc_hbm_fetch -name '*counter_collection.csv' | head -1)
W=$(find $ROOT/gpurun_out/pmc_hbm_write -name '*counter_collection.csv' | head -1)
cd $ROOT
python3 - "$F" "$W" <<'PY' | tee $ROOT/gpurun_out/r03_pmc_hbm_kernels_B512.txt
import collections, csv, os, re, sys
sys.path.insert(0, os.getcwd())
from eeggan_amd import _lib
print(f"# library {_lib.library_id()}")
print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 scripts/bench_hbm.py 512")
print("# mean per launch over the timed launches of each kernel; read = 2 x FETCH_SIZE KiB (gfx950 tallies a 128-byte")
print("# read at 64 bytes), written = WRITE_SIZE KiB; tensors [512, 64, 1920] float32 (251.7 MB), resampler rows 1280 / 1920 / 1024 long")
acc = collections.OrderedDict()
for tag, path in (("fetch", sys.argv[1]), ("write", sys.argv[2])):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "eeggan" not in n:
            continue
        k = re.sub(r"eeggan::\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
        key = (k, r.get("Grid_Size", ""))
        acc.setdefault(key, {}).setdefault(tag, []).append(float(r["Counter_Value"]))
for (k, grid), v in acc.items():
    f, w = v.get("fetch", [0.0]), v.get("write", [0.0])
    f, w = f[len(f) // 3:], w[len(w) // 3:]
    rd, wr = 2 * sum(f) / len(f) * 1024 / 1e6, sum(w) / len(w) * 1024 / 1e6
    if rd + wr < 50:
        continue
    print(f"{k:58s} grid {grid:>9s}  n={len(f):3d}  read {rd:8.1f} MB  written {wr:8.1f} MB  total {rd + wr:8.1f} MB")
PY
