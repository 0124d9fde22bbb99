#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3ac; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 300 python -m pytest tests/test_gpu_fused_block.py -q -k "wgrad16zs" -x > $O/t.log 2>&1; rc=$?; echo "wgrad16zs tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t.log | tail -8 | cut -c1-300
[ $rc -eq 0 ] || exit 1
for v in 1 0; do
  MVD_WGRAD16ZS=$v run timeout -k 10 300 python tools/bench_conv.py --dtype bf16 --layers enc1.conv0,enc2.conv0 --what wgrad --iters 20 > $O/conv_$v.log 2>&1; echo "--- WGRAD16ZS=$v"; grep -v amdgpu $O/conv_$v.log
done
