#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
ROOT=$PWD
O=$PWD/gpurun_out/r3l; mkdir -p $O
cd /tmp
for v in 1 0; do
  MVD_WGRAD16Z=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$v -- python3 $ROOT/tools/bench_conv.py --dtype bf16 --layers ${LAYERS:-enc0.conv1,dec5.conv0,enc1.conv1,enc2.conv1} --what wgrad --iters 20 > $O/conv_z$v.log 2>&1
  cp $(ls $O/p$v/*/*kernel_stats.csv | head -1) $O/stats_z$v.csv; rm -rf $O/p$v
  echo "--- Z=$v"; cut -d, -f1-7 $O/stats_z$v.csv | cut -c1-160 | head -12
done
