#!/bin/bash
# rocprofv3 kernel statistics of the bench step (run through gpurun from the repo root); ARGS = extra bench.py flags
set -e
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_${TAG:-fp32}
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline $ARGS > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log
