#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
ROOT=$PWD
O=$PWD/gpurun_out/r3q; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_bf16 -- python3 $ROOT/bench.py --precision bf16 --no-graph --steps 4 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/trace_bf16.log 2>&1
python3 $ROOT/tools/trace_top.py $O/trace_bf16 70 > $O/bf16_step_per_launch.txt 2>&1
rm -rf $O/trace_bf16
head -75 $O/bf16_step_per_launch.txt
