"""Rows of the stand-alone weight-stationary launch (64 -> 64 channels, k = 3, T = 1920, 32 samples) from the two
rocprofv3 counter passes (one per counter, each its own run of ``scripts/bench_one.py 32 64 64 1920 3``), in the
form ``bench.py`` reads for ``roofline.traffic``.

   python scripts/pmc_conv_ws.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/<file>.csv

Values are KiB per dispatch as rocprofv3 reports them; on gfx950 FETCH_SIZE tallies a 128-byte read request at
64 bytes, so readers double it."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import _lib  # noqa: E402

print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 scripts/bench_one.py 32 64 64 1920 3")
print("# units: KiB per dispatch as reported; FETCH_SIZE is tallied at half for 128-byte requests on gfx950 (double it)")
print(f"# library {_lib.library_id()}")
print("pass,dispatch,kernel,counter,value_kib,duration_ns")
for tag, path in (("fetch", sys.argv[1]), ("write", sys.argv[2])):
    rows = [r for r in csv.DictReader(open(path)) if "conv_ws_kernel" in r["Kernel_Name"]]
    rows = rows[len(rows) // 5:]  # the first launches warm the caches
    for r in rows:
        name = r["Kernel_Name"].replace("eeggan::(anonymous namespace)::", "").split("(")[0]
        dur = int(r.get("End_Timestamp", 0) or 0) - int(r.get("Start_Timestamp", 0) or 0)
        print(f'{tag},{r.get("Dispatch_Id", "")},"{name}",{r["Counter_Name"]},{float(r["Counter_Value"]):.6f},{dur}')
