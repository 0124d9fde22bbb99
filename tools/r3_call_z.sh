#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3z; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 300 python -m pytest tests/test_gpu_fused_block.py -q -k "stride2" -x > $O/t.log 2>&1; rc=$?; echo "stride2 tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t.log | tail -8
[ $rc -eq 0 ] || exit 1
for v in 1 0; do
  MVD_DGRAD16SP=$v run timeout -k 10 300 python tools/bench_conv.py --dtype bf16 --layers enc1.conv0,enc2.conv0 --what dgrad --iters 20 > $O/conv_$v.log 2>&1; echo "--- DGRAD16SP=$v"; grep -v amdgpu $O/conv_$v.log
done
for v in 1 0 1 0; do
  MVD_DGRAD16SP=$v run timeout -k 10 300 python bench.py --precision bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $O/b_$v.json 2> $O/e; echo "bf16 SP=$v: $(python -c "import json; d=json.loads(open('$O/b_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
