#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3j; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_bf16.py -q > $O/t1.log 2>&1; echo "fused+bf16 rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t1.log | tail -12
run timeout -k 10 600 python -m pytest tests/test_gpu_cfg2.py -q -k "bf16" > $O/t2.log 2>&1; echo "cfg2 bf16 rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/t2.log | tail -5
for v in 1 0; do
  MVD_DGRAD16S=$v run timeout -k 10 300 python tools/bench_conv.py --dtype bf16 --layers enc1.conv0,enc2.conv0 --what dgrad --iters 20 > $O/conv_s$v.log 2>&1; echo "--- DGRAD16S=$v"; grep -v amdgpu $O/conv_s$v.log
  MVD_DGRAD16S=$v run timeout -k 10 300 python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_bf16_$v.json 2> $O/e; echo "bf16: $(python -c "import json; d=json.loads(open('$O/b_bf16_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
echo done
