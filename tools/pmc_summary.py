"""Summarises the rocprofv3 --pmc passes of tools/pmc_kernel.sh: per kernel name, the mean of every counter over the
dispatches of the LARGEST-grid kernel family (the benchmarked layer), plus mean duration from the kernel traces.
usage: python tools/pmc_summary.py gpurun_out/pmc_TAG [name-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "mvd::"
    counters = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(list)
    for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if filt in name:
                counters[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(root, "p*", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if filt in name:
                durs[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name in sorted(counters, key=lambda n: -sum(durs.get(n, [0]))):
        d = durs.get(name, [])
        d2 = sorted(d)[len(d) // 2] if d else float("nan")
        print(f"== {name[:110]}\n   dispatches {len(d)}  median duration {d2:.1f} us (under the profiler)")
        for c, v in sorted(counters[name].items()):
            print(f"   {c:28s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")


if __name__ == "__main__":
    main()
