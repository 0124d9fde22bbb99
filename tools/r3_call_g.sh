#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3g; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
run timeout -k 10 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_ddp.py -q > $O/t1.log 2>&1; echo "graph+ddp rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/t1.log | tail -12
for sh in 1 0; do
  MVD_SHARE_GRADS=$sh run timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/b_fp32_s$sh.json 2>$O/e
  MVD_SHARE_GRADS=$sh run timeout -k 10 300 python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_bf16_s$sh.json 2>$O/e
  echo "share=$sh: fp32 $(python -c "import json; d=json.loads(open('$O/b_fp32_s$sh.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])") bf16 $(python -c "import json; d=json.loads(open('$O/b_bf16_s$sh.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
run timeout -k 10 400 python tools/ddp_overlap_probe.py overlap fp32 $O/ddp_overlap_fp32.json > $O/ddp1.log 2>&1; echo "overlap fp32 rc=$?"; tail -3 $O/ddp1.log
run timeout -k 10 400 python tools/ddp_overlap_probe.py overlap bf16 $O/ddp_overlap_bf16.json > $O/ddp2.log 2>&1; echo "overlap bf16 rc=$?"; tail -3 $O/ddp2.log
run timeout -k 10 300 python tools/ddp_overlap_probe.py hooks fp32 > $O/hooks_fp32.json 2>$O/e; cat $O/hooks_fp32.json
run timeout -k 10 300 python tools/ddp_overlap_probe.py hooks bf16 > $O/hooks_bf16.json 2>$O/e; cat $O/hooks_bf16.json
echo done
