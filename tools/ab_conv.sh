#!/bin/bash
# usage: tools/ab_conv.sh <variant .so> [bench_conv.py args]: tools/bench_conv.py alternately on the in-tree library (A) and
# a variant library (B), twice each, on one box
V=$1; shift
for r in 1 2; do
  echo "--- A (tree)"; python tools/bench_conv.py "$@" 2>&1 | grep -v amdgpu.ids
  echo "--- B (variant)"; bash tools/run_with_lib.sh $V python tools/bench_conv.py "$@" 2>&1 | grep -v amdgpu.ids
done
