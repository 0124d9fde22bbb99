#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3t; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
for v in 1 0; do
  MVD_FWD16_S2=$v run timeout -k 10 300 python tools/bench_conv.py --dtype bf16 --layers enc1.conv0,enc2.conv0 --what fwd --iters 20 > $O/conv_s$v.log 2>&1; echo "--- FWD16_S2=$v"; grep -v amdgpu $O/conv_s$v.log
done
run timeout -k 10 600 python -m pytest tests/test_gpu_cfg2.py tests/test_gpu_bf16.py -q -k "bf16" > $O/t2.log 2>&1; echo "cfg2+bf16 rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/t2.log | tail -8
for v in 1 0; do
  MVD_FWD16_S2=$v run timeout -k 10 300 python bench.py --precision bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $O/b_$v.json 2> $O/e; echo "bf16 S2=$v: $(python -c "import json; d=json.loads(open('$O/b_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
