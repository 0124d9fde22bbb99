#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3s; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/all.log 2>&1; echo "all gpu tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/all.log | tail -15
run timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 3000 $O/bench_default.json
