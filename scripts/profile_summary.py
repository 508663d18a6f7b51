"""Per-kernel totals of a rocprofv3 kernel-trace CSV, normalised per training step.
   python scripts/profile_summary.py <kernel_trace.csv> <steps-equivalents>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    name = re.sub(r"eeggan::\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:72]
    agg[name][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[name][1] += 1
print(f"launches/step {len(rows) / steps:.1f}   kernel time/step {sum(v[0] for v in agg.values()) / steps / 1e6:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:74s} n/step={v[1] / steps:7.1f} ms/step={v[0] / steps / 1e6:7.3f} avg={v[0] / v[1] / 1e3:7.1f}us")
