#!/bin/bash
# shader clock under load: GRBM_GUI_ACTIVE cycles / kernel duration (run through gpurun from the repo root)
set -e
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
OUT=$R/gpurun_out/pmc_clock
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/bench_conv.py --layers ${LAYER:-dec5.conv0,enc0.conv1} --what ${WHAT:-fwd,wgrad} --iters 3 > $OUT/log 2>&1
echo done
