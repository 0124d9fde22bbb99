#!/bin/bash
# usage (on the GPU box): tools/prof_step.sh <tag> [bench.py args]: kernel-trace statistics of 8 bench steps ->
# gpurun_out/<tag>_kernel_stats.csv (+ the 25 most expensive kernels on stdout)
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/$TAG.log 2>&1 || { echo "$TAG failed"; tail -5 $OUT/$TAG.log; exit 1; }
F=$(ls /tmp/prof_$TAG/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -z "$F" ] && { echo "no stats file"; exit 1; }
cp "$F" $OUT/${TAG}_kernel_stats.csv
mkdir -p $OUT/${TAG}_trace && cp "${F%kernel_stats.csv}kernel_trace.csv" $OUT/${TAG}_trace/ 2>/dev/null || true
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print(f'{float(r["TotalDurationNs"])/11e3:9.1f} us/step  x{int(r["Calls"])//11:<3d} {r["Name"][:90]}')
PY
