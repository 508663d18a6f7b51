"""Per-kernel totals of a rocprofv3 kernel-trace CSV of `bench.py`, per training step.

   python scripts/profile_summary.py <kernel_trace.csv> [steps]

Only the last `steps` (default 10) training steps of the trace are counted: a step ends with its sixth
Adam launch (five critic updates and one generator update), so the window runs from the Adam launch that
closes the step before them to the one that closes the last; warm-up, graph capture and the stand-alone
kernel timing that follows the timed region stay outside.  The launches of the weight-stationary conv kernel
right AFTER the last optimizer launch (the first 105) are that stand-alone timing (`bench.py`'s roofline figure, one
shape launched back to back); their mean duration is printed for comparison with the HIP-event figure in the bench
line.  (The family's chain of a whole step, which `bench.py` replays after them, follows and is not counted.)"""
import collections
import csv
import os
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
if len(adam) < 6 * steps + 1:
    sys.exit(f"trace holds {len(adam)} optimizer launches, fewer than {steps} steps")
first, last = adam[-6 * steps - 1] + 1, adam[-1]
window = rows[first:last + 1]
wall = int(window[-1]["End_Timestamp"]) - int(rows[first - 1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
for r in window:
    name = re.sub(r"eeggan::\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:72]
    agg[name][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[name][1] += 1
total = sum(v[0] for v in agg.values())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeggan_amd import _lib  # noqa: E402
print(f"# library {_lib.library_id()}  (the build of libeeggan_hip.so this trace ran)")
print(f"last {steps} steps: launches/step {len(window) / steps:.1f}   kernel time/step {total / steps / 1e6:.3f} ms   "
      f"wall/step {wall / steps / 1e6:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:74s} n/step={v[1] / steps:7.1f} ms/step={v[0] / steps / 1e6:7.3f} ({100.0 * v[0] / total:4.1f}%) "
          f"avg={v[0] / v[1] / 1e3:7.1f}us")
tail = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[last + 1:] if "conv_ws_kernel" in r["Kernel_Name"]]
tail = tail[:105]  # 5 warm-up launches and two timed runs of 50; what follows is the family's chain (all layer sizes)
if tail:
    timed = tail[len(tail) // 5:]  # the first launches are the warm-up of that loop
    print(f"stand-alone roofline launches after the timed steps: {len(tail)} of conv_ws_kernel, mean of the last "
          f"{len(timed)}: {sum(timed) / len(timed) / 1e3:.2f} us")
