"""Per-launch view of a rocprofv3 --kernel-trace CSV: the launches of the LAST complete train step (between two
k_sgd launches), largest first, and the per-kernel-name totals of that step.
usage: python tools/trace_top.py <dir with *kernel_trace.csv> [n]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = max(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)  # newest
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
sgd = [i for i, r in enumerate(rows) if "k_sgd" in r[2]]
if len(sgd) >= 2:
    rows = rows[sgd[-2] + 1: sgd[-1] + 1]
tot = sum(e - s for s, e, _ in rows)
print(f"{len(rows)} launches, sum of kernel durations {tot / 1e6:.3f} ms, span {(rows[-1][1] - rows[0][0]) / 1e6:.3f} ms")
by = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    key = n.split("(")[0][-70:]
    by[key][0] += e - s
    by[key][1] += 1
print("--- by kernel")
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{t / 1e3:9.1f} us  x{c:<4d} {k}")
print("--- largest launches")
for s, e, n in sorted(rows, key=lambda r: r[0] - r[1])[:top]:
    print(f"{(e - s) / 1e3:9.1f} us  {n.split('(')[0][-70:]}")
