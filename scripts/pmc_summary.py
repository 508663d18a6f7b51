"""Mean per-dispatch value of one rocprofv3 counter per kernel name.

   python scripts/pmc_summary.py <counter_collection.csv> [substring ...]

FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  On gfx950 FETCH_SIZE tallies 128-byte
read requests at 64 bytes: the printed figure for it is already doubled (``x2``)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    name = re.sub(r"eeggan::\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:72]
    if want and not any(w in name for w in want):
        continue
    key = (name, r["Counter_Name"])
    agg[key][0] += float(r["Counter_Value"])
    agg[key][1] += 1
for (name, counter), (total, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    mean = total / n
    if counter == "FETCH_SIZE":
        print(f"{name:74s} {counter:10s} n={n:6d} mean={2 * mean:12.1f} KiB/dispatch (x2)")
    else:
        print(f"{name:74s} {counter:10s} n={n:6d} mean={mean:12.1f} KiB/dispatch")
