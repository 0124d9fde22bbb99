#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3af; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_fused_block.py -q -x -k "seghead or seg_head" > $O/t.log 2>&1; rc=$?; echo "seghead tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t.log | tail -8 | cut -c1-300
[ $rc -eq 0 ] || exit 1
bash tools/r3_call_ae.sh
