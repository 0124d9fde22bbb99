#!/bin/bash
# GRBM_GUI_ACTIVE (cycles, clock-independent) + kernel-trace duration of one bench_conv layer under env switches.
#   LAYER=enc0.conv1 WHAT=dgrad DTYPE=bf16 TAG=x tools/pmc_cycles.sh     (through gpurun from the repo root)
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
OUT=$R/gpurun_out/cyc_${TAG:-x}
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/bench_conv.py --layers ${LAYER:-enc0.conv1} --what ${WHAT:-dgrad} --dtype ${DTYPE:-bf16} --iters 6 > $OUT/p1.log 2>&1 || echo failed
python3 $R/tools/pmc_summary.py $OUT ${FILTER:-k_fwd16} | grep -v "^   SQ_BUSY"
