#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3ad; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_fused_block.py -q -x > $O/t.log 2>&1; rc=$?; echo "fused tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t.log | tail -8 | cut -c1-300
[ $rc -eq 0 ] || exit 1
for m in 1 0 1 0; do
  MVD_FUSE_SEGHEAD=$m run timeout -k 10 300 python bench.py --precision bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $O/b_$m.json 2> $O/e; echo "bf16 seghead-fuse=$m: $(python -c "import json; d=json.loads(open('$O/b_$m.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
