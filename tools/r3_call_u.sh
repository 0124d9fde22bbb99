#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3u; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
for v in 1 0; do
  MVD_X=$v run timeout -k 10 300 python tools/bench_conv.py --dtype bf16 --layers enc2.conv1,dec3.conv0 --what fwd,dgrad --iters 30 > $O/conv_$v.log 2>&1; echo "--- run=$v"; grep -v amdgpu $O/conv_$v.log
done
