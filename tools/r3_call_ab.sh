#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3ab; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_cfg2.py tests/test_gpu_bf16.py tests/test_gpu_fused_block.py tests/test_gpu_graph.py -q -k "bf16 or fused or fwd16 or graph" > $O/t.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/t.log | tail -6
for v in 1 0 1 0; do
  MVD_WGRAD16ZS=$v run timeout -k 10 300 python bench.py --precision bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $O/b_$v.json 2> $O/e; echo "bf16 ZS=$v: $(python -c "import json; d=json.loads(open('$O/b_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
