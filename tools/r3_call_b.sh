#!/bin/bash
# round-3 GPU call B: fused block (tests, piece timings), graph tests, regression of the bf16 suite, bench A/B
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3b; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_fused_block.py -q > $O/fused_tests.log 2>&1; echo "fused tests rc=$?"; tail -15 $O/fused_tests.log
run timeout -k 10 300 python tools/bench_fused_block.py > $O/fused_block.json 2> $O/fused_block.err; echo "fused bench rc=$?"; cat $O/fused_block.json
run timeout -k 10 600 python -m pytest tests/test_gpu_graph.py -q > $O/graph_tests.log 2>&1; echo "graph tests rc=$?"; tail -5 $O/graph_tests.log
run timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -q -x > $O/bf16_tests.log 2>&1; echo "bf16 tests rc=$?"; tail -5 $O/bf16_tests.log
for v in "1" "0"; do
  MVD_BF16_CONV_STATS=$v run timeout -k 10 300 python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_bf16_stats$v.json 2> $O/b_bf16_stats$v.err
  echo "bf16 stats=$v: $(python -c "import json; d=json.loads(open('$O/b_bf16_stats$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
done
MVD_FUSE_PROLOGUE_TRAIN=1 run timeout -k 10 300 python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_bf16_protrain.json 2> $O/b_bf16_protrain.err
echo "bf16 prologue in training: $(python -c "import json; d=json.loads(open('$O/b_bf16_protrain.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
echo done
