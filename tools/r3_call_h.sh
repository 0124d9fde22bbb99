#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
O=$PWD/gpurun_out/r3h; mkdir -p $O
run() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi; return $rc; }
R=$PWD
cd /tmp
for tag in bf16 fp32; do
  extra=""; [ $tag = bf16 ] && extra="--precision bf16"
  run timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tag -- python3 $R/bench.py $extra --no-graph --steps 4 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/trace_$tag.log 2>&1
  python3 $R/tools/trace_top.py $O/trace_$tag 60 > $O/r03_${tag}_step_per_launch.txt 2>&1; head -2 $O/r03_${tag}_step_per_launch.txt
  rm -rf $O/trace_$tag
done
cd $R
run timeout -k 10 400 python tools/ddp_overlap_probe.py overlap bf16 $O/ddp_overlap_bf16.json > $O/ddp2.log 2>&1; echo "overlap bf16 rc=$?"
python -c "
import json; d=json.load(open('$O/ddp_overlap_bf16.json')); print({k:v for k,v in d.items() if k!='buckets'}); [print(b) for b in d['buckets']]"
echo done
