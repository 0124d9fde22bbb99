#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3v; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_cfg2.py tests/test_gpu_bf16.py -q -k "bf16" > $O/t2.log 2>&1; echo "cfg2+bf16 rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/t2.log | tail -8
run timeout -k 10 300 python bench.py --precision bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $O/b.json 2> $O/e; echo "bf16: $(python -c "import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
