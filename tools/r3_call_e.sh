#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3e; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_fused_block.py -q > $O/t1.log 2>&1; echo "fused rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/t1.log | tail -8
run timeout -k 10 300 python tools/bench_zmarch.py > $O/zmarch.json 2> $O/zmarch.err; echo "zmarch rc=$?"; cat $O/zmarch.json
run timeout -k 10 300 python tools/bench_fused_block.py > $O/fused_block.json 2> $O/fused_block.err; echo "fused bench rc=$?"; cat $O/fused_block.json
run timeout -k 10 600 python -m pytest tests/test_gpu_cfg2.py -q -k "gradients_vs_fp64 or bf16" > $O/t2.log 2>&1; echo "cfg2 subset rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/t2.log | tail -5
run timeout -k 10 300 python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/b_bf16.json 2> $O/b_bf16.err; echo "bf16: $(python -c "import json; d=json.loads(open('$O/b_bf16.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], 'block', r['block_ms'], r['frac'], 'conv', r['conv_only'], 'inf', r.get('inference_form'))" 2>&1 | tail -1)"
echo done
