#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3i; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_fused_block.py -q > $O/t1.log 2>&1; echo "fused rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/t1.log | tail -12
for v in 1 0; do
  MVD_FWD16Y=$v run timeout -k 10 300 python tools/bench_conv.py --dtype bf16 --layers enc1.conv1,dec4.conv0 --iters 20 > $O/conv_y$v.log 2>&1; echo "--- FWD16Y=$v"; cat $O/conv_y$v.log | grep -v amdgpu
done
run timeout -k 10 300 python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_bf16.json 2> $O/e; echo "bf16: $(python -c "import json; d=json.loads(open('$O/b_bf16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
echo done
