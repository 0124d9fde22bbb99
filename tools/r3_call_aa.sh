#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3aa; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 300 python -m pytest tests/test_gpu_fused_block.py -q -k "fwd16ys" -x > $O/t.log 2>&1; rc=$?; echo "fwd16ys tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t.log | tail -8
[ $rc -eq 0 ] || exit 1
for v in 1 0; do
  MVD_FWD16YS=$v run timeout -k 10 300 python tools/bench_conv.py --dtype bf16 --layers enc1.conv0 --what fwd --iters 20 > $O/conv_$v.log 2>&1; echo "--- FWD16YS=$v"; grep -v amdgpu $O/conv_$v.log
done
