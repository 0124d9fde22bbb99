#!/bin/bash
# instruction-cache counters of the conv kernels (run through gpurun from the repo root)
set -e
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
OUT=$R/gpurun_out/pmc_icache
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/bench_conv.py --layers ${LAYER:-dec5.conv0} --what ${WHAT:-fwd,wgrad} --iters 2 > $OUT/p1.log 2>&1 || tail -5 $OUT/p1.log
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/tools/bench_conv.py --layers ${LAYER:-dec5.conv0} --what ${WHAT:-fwd,wgrad} --iters 2 > $OUT/p2.log 2>&1 || tail -5 $OUT/p2.log
echo done
