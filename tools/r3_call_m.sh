#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3m; mkdir -p $O
for d in ${DBGS:-0 1 2 8 3 16}; do
  MVD_WG16Z_DBG=$d timeout -k 10 120 python tools/bench_conv.py --dtype bf16 --layers enc0.conv1 --what wgrad --iters 20 2>&1 | grep -v amdgpu | sed "s/^/DBG=$d /"
done
