#!/bin/bash
# round-3 GPU call A: hipGraph step (tests + A/B on every config), the new bench line, a per-launch trace of the graphed bf16 step
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3a; mkdir -p $O
run() {  # a step that timed out / was killed ends the call (no further GPU step after a kill)
  "$@"; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi
  return $rc
}
run timeout -k 10 600 python -m pytest tests/test_gpu_graph.py -x -q > $O/graph_tests.log 2>&1; echo "graph tests rc=$?"; tail -5 $O/graph_tests.log
run timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; cut -c1-600 $O/bench_default.json
for cfg in "--precision bf16" "--config cfg3" "--config cfg4" "--config cfg5"; do
  for g in "" "--no-graph"; do
    tag=$(echo "$cfg$g" | tr -d ' -')
    run timeout -k 10 300 python bench.py $cfg $g --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_$tag.json 2> $O/b_$tag.err
    echo "$cfg $g rc=$? $(python -c "import json,sys; d=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['step_launch'][:20])" 2>&1 | tail -1)"
  done
done
run timeout -k 10 300 python bench.py --no-graph --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/b_fp32_nograph.json 2> $O/b_fp32_nograph.err
echo "fp32 eager: $(cut -c1-200 $O/b_fp32_nograph.json)"
cd /tmp
run timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_bf16 -- python3 $OLDPWD/bench.py --precision bf16 --steps 4 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_bf16.log 2>&1
cd $OLDPWD
python tools/trace_top.py $O/trace_bf16 45 > $O/r03_bf16_graph_step_per_launch.txt 2>&1; head -3 $O/r03_bf16_graph_step_per_launch.txt
rm -rf $O/trace_bf16
echo done
