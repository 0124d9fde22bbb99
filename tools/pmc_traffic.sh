#!/bin/bash
# HBM traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in separate passes (run through gpurun from the repo root)
set -e
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
OUT=$R/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/tools/bench_conv.py --layers dec5.conv0 --what fwd --iters 2 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/tools/bench_conv.py --layers dec5.conv0 --what fwd --iters 2 > $OUT/write.log 2>&1
echo done
