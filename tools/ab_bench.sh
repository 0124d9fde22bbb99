#!/bin/bash
# usage: tools/ab_bench.sh <variant .so> [bench.py args]: alternates `python bench.py ...` between the in-tree library (A)
# and a variant library (B) on the same box, twice each, and prints value / ms_per_step -- box-to-box variation (3-5 %)
# is larger than most single-kernel effects, so before/after numbers are only comparable inside one call.
V=$1; shift
for r in 1 2; do
  echo -n "A (tree)    : "; python bench.py "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
  echo -n "B (variant) : "; bash tools/run_with_lib.sh $V python bench.py "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
done
