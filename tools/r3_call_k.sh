#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3k; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 300 python -m pytest tests/test_gpu_fused_block.py -q -k "wgrad16z" -x > $O/t1.log 2>&1; rc=$?; echo "wgrad16z tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t1.log | tail -12
[ $rc -eq 0 ] || exit 1
for v in 1 0; do
  MVD_WGRAD16Z=$v run timeout -k 10 300 python tools/bench_conv.py --dtype bf16 --layers enc0.conv1,dec5.conv0,enc1.conv1,dec4.conv0,enc2.conv1,dec3.conv0 --what wgrad --iters 20 > $O/conv_z$v.log 2>&1; echo "--- WGRAD16Z=$v"; grep -v amdgpu $O/conv_z$v.log
done
run timeout -k 10 600 python -m pytest tests/test_gpu_cfg2.py tests/test_gpu_bf16.py -q -k "bf16" > $O/t2.log 2>&1; echo "cfg2+bf16 rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/t2.log | tail -8
for v in 1 0; do
  MVD_WGRAD16Z=$v run timeout -k 10 300 python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_bf16_$v.json 2> $O/e; echo "bf16 Z=$v: $(python -c "import json; d=json.loads(open('$O/b_bf16_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
echo done
