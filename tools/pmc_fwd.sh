#!/bin/bash
# PMC passes over the forward Winograd kernel on the dominant layer (run through gpurun from the repo root)
set -e
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
OUT=$R/gpurun_out/pmc_fwd
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/bench_conv.py --layers ${LAYER:-dec5.conv0} --what ${WHAT:-fwd} --iters 2 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/tools/bench_conv.py --layers ${LAYER:-dec5.conv0} --what ${WHAT:-fwd} --iters 2 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/p3 -- python3 $R/tools/bench_conv.py --layers ${LAYER:-dec5.conv0} --what ${WHAT:-fwd} --iters 2 > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/p4 -- python3 $R/tools/bench_conv.py --layers ${LAYER:-dec5.conv0} --what ${WHAT:-fwd} --iters 2 > $OUT/p4.log 2>&1
echo done
