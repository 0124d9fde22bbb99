#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3f; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/all.log 2>&1; echo "all gpu tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/all.log | tail -12
echo done
