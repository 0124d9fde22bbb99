#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3p; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
for ng in 2 1 2 1; do
  MVD_WGRAD16Z_NG=$ng run timeout -k 10 300 python bench.py --precision bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $O/b_bf16_$ng.json 2> $O/e; echo "bf16 NG=$ng: $(python -c "import json; d=json.loads(open('$O/b_bf16_$ng.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
MVD_WGRAD16Z=0 run timeout -k 10 300 python bench.py --precision bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $O/b_bf16_old.json 2> $O/e; echo "bf16 old: $(python -c "import json; d=json.loads(open('$O/b_bf16_old.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
