#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3w; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_cfg2.py tests/test_gpu_bf16.py -q -k "wgrad or bf16 or step" -x > $O/t.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t.log | tail -8
[ $rc -eq 0 ] || exit 1
for v in 1 0 1 0; do
  MVD_WGRAD_REDUCE_G16=$v run timeout -k 10 300 python bench.py --precision bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $O/b_$v.json 2> $O/e; echo "bf16 G16=$v: $(python -c "import json; d=json.loads(open('$O/b_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
for v in 1 0; do
  MVD_WGRAD_REDUCE_G16=$v run timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/f_$v.json 2> $O/e; echo "fp32 G16=$v: $(python -c "import json; d=json.loads(open('$O/f_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
