#!/bin/bash
# PMC passes (each in its own rocprofv3 run, --kernel-trace only) over one layer / pass of tools/bench_conv.py.
#   LAYER=dec5.conv1 WHAT=fwd DTYPE=bf16 TAG=fwd16q tools/pmc_kernel.sh      (run through gpurun from the repo root)
# Output: gpurun_out/pmc_$TAG/p*/ ... counter_collection.csv + kernel_trace.csv; summarise with tools/pmc_summary.py
set -e
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
TAG=${TAG:-k}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="--layers ${LAYER:-dec5.conv1} --what ${WHAT:-fwd} --dtype ${DTYPE:-fp32} --iters ${ITERS:-3}"
cd /tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_conv.py $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
