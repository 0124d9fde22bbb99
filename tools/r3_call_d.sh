#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3d; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 900 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_bf16.py tests/test_gpu_graph.py -q > $O/t1.log 2>&1; echo "fused+bf16+graph rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/t1.log | tail -15
run timeout -k 10 1100 python -m pytest tests/test_gpu_cfg2.py tests/test_gpu_configs.py -q > $O/t2.log 2>&1; echo "cfg2+configs rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|\[cfg" $O/t2.log | tail -15
for v in 1 0; do
  MVD_FWD16Y=$v run timeout -k 10 300 python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_bf16_y$v.json 2> $O/b_bf16_y$v.err
  echo "bf16 y=$v: $(python -c "import json; d=json.loads(open('$O/b_bf16_y$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
done
run timeout -k 10 300 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_cfg4.json 2> $O/b_cfg4.err; echo "cfg4: $(cut -c1-160 $O/b_cfg4.json)"
run timeout -k 10 300 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_cfg5.json 2> $O/b_cfg5.err; echo "cfg5: $(cut -c1-160 $O/b_cfg5.json)"
echo done
