#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3n; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
for ng in 2 1; do
MVD_WGRAD16Z_NG=$ng run timeout -k 10 300 python -m pytest tests/test_gpu_fused_block.py -q -k "wgrad16z" -x > $O/t$ng.log 2>&1; rc=$?; echo "NG=$ng wgrad16z tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t$ng.log | tail -12
[ $rc -eq 0 ] || exit 1
done
for ng in 2 1; do
  export MVD_WGRAD16Z_NG=$ng
  echo "== NG=$ng"
  DBGS="0 128 2 130 3 131" bash tools/r3_call_m.sh
done
