#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3ae; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
for m in 1 0 1 0 1 0; do
  MVD_FUSE_SEGHEAD=$m run timeout -k 10 300 python bench.py --steps 40 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline > $O/b_$m.json 2> $O/e; echo "bf16 seghead-fuse=$m: $(python -c "import json; d=json.loads(open('$O/b_$m.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
