"""Summarise a rocprofv3 kernel-trace CSV: mean duration per (kernel, grid size)."""
import csv, sys, collections, re
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"eeggan::\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)[:60]
        g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]))
        rows[(name, g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = sorted(rows.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in rows.values())
print(f"total kernel time {tot/1e6:.2f} ms")
for (name, g), v in out[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{name:62s} grid={str(g):14s} n={len(v):5d} mean={sum(v)/len(v)/1e3:8.1f}us min={min(v)/1e3:8.1f} total={sum(v)/1e6:8.2f}ms")

if len(sys.argv) > 3 and sys.argv[3] == "seq":
    print("--- consecutive runs ---")
    runs = []
    with open(sys.argv[1]) as f:
        rows_ = sorted(csv.DictReader(f), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows_:
        name = re.sub(r"eeggan::\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)[:48]
        g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]))
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if runs and runs[-1][0] == (name, g):
            runs[-1][1].append(d)
        else:
            runs.append(((name, g), [d]))
    for (name, g), v in runs:
        if len(v) >= 10:
            v2 = v[3:]
            print(f"{name:50s} grid={str(g):12s} n={len(v):4d} mean={sum(v2)/len(v2)/1e3:8.1f}us min={min(v2)/1e3:8.1f}")
