#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3n; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 300 python -m pytest tests/test_gpu_fused_block.py -q -k "wgrad16z" -x > $O/t1.log 2>&1; rc=$?; echo "wgrad16z tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t1.log | tail -12
[ $rc -eq 0 ] || exit 1
bash tools/r3_call_m.sh
