#!/bin/bash
# Round-3 profile collection (one gpurun call from the repo root).  Output: gpurun_out/profiles_r03/ ; the summaries are copied
# into profiles/ by hand (this script does not touch profiles/).
#   * rocprofv3 --kernel-trace --stats of the bench commands (fp32 default, bf16, cfg3, cfg4, cfg5), eager AND graphed
#   * per-launch listing of one step (tools/trace_top.py) for fp32 / bf16
#   * the isolated roofline launches (--roofline-only)
#   * PMC passes (each counter set in its own run, --kernel-trace only) of k_fwd16y on 32 -> 32 and 32+32 -> 32 at 128^3,
#     k_dgrad16s, and the fp32 strided kernels
set -e
export TMPDIR=/tmp PYTHONPATH=$PWD
ROOT=$PWD
RR=r03
OUT=$ROOT/gpurun_out/profiles_$RR
mkdir -p $OUT
cd /tmp
stats() {  # tag, bench args...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary "$@" > $OUT/$tag.log 2>&1 || echo "$tag failed"
  cp $(ls $OUT/$tag/*/*kernel_stats.csv | head -1) $OUT/${RR}_${tag}_kernel_stats.csv
  tail -1 $OUT/$tag.log | cut -c1-300
  rm -rf $OUT/$tag
}
# the bench lines themselves, without the profiler (fp32 default line incl. secondary.bf16; the other BASELINE configs)
plain() {  # tag, bench args...
  tag=$1; shift
  python3 $ROOT/bench.py "$@" > $OUT/plain_$tag.json 2> $OUT/plain_$tag.err || echo "plain $tag failed"
  tail -1 $OUT/plain_$tag.json | cut -c1-200
}
plain default --gpus 1 --steps 20 --warmup 5
plain bf16 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline
plain cfg3 --config cfg3 --steps 10 --warmup 5 --no-cpu-baseline --no-roofline
plain cfg4 --config cfg4 --steps 10 --warmup 5 --no-cpu-baseline --no-roofline
plain cfg5 --config cfg5 --steps 10 --warmup 5 --no-cpu-baseline --no-roofline
python3 - <<PY
import json
out = {}
for t in ("default", "bf16", "cfg3", "cfg4", "cfg5"):
    try:
        out[t] = json.loads(open("$OUT/plain_%s.json" % t).read().strip().splitlines()[-1])
    except Exception as e:
        out[t] = {"error": str(e)}
json.dump(out, open("$OUT/${RR}_bench_lines.json", "w"), indent=1)
PY
stats bench
stats bench_bf16 --precision bf16
stats bench_cfg3 --config cfg3
stats bench_cfg4 --config cfg4
stats bench_cfg5 --config cfg5
for tag in fp32 bf16; do
  extra=""; [ $tag = bf16 ] && extra="--precision bf16"
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$tag -- python3 $ROOT/bench.py $extra --no-graph --steps 4 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $OUT/trace_$tag.log 2>&1
  python3 $ROOT/tools/trace_top.py $OUT/trace_$tag 60 > $OUT/${RR}_${tag}_step_per_launch.txt 2>&1
  rm -rf $OUT/trace_$tag
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roof_fp32 -- python3 $ROOT/bench.py --roofline-only > $OUT/roof_fp32.log 2>&1
cp $(ls $OUT/roof_fp32/*/*kernel_stats.csv | head -1) $OUT/${RR}_roofline_fp32_kernel_stats.csv; rm -rf $OUT/roof_fp32
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roof_bf16 -- python3 $ROOT/bench.py --roofline-only --precision bf16 > $OUT/roof_bf16.log 2>&1
cp $(ls $OUT/roof_bf16/*/*kernel_stats.csv | head -1) $OUT/${RR}_roofline_bf16_kernel_stats.csv; rm -rf $OUT/roof_bf16
tail -1 $OUT/roof_fp32.log | cut -c1-400; tail -1 $OUT/roof_bf16.log | cut -c1-600
cd $ROOT
pmc() {  # tag layer what dtype
  LAYER=$2 WHAT=$3 DTYPE=$4 ITERS=10 TAG=${RR}_$1 tools/pmc_kernel.sh > $OUT/pmc_$1.txt 2>&1 || true
  cp gpurun_out/pmc_${RR}_$1/summary.txt $OUT/${RR}_pmc_$1_summary.txt || true
  rm -rf gpurun_out/pmc_${RR}_$1
}
pmc fwd16y_32_32 dec5.conv1 fwd bf16
pmc fwd16y_64_32 dec5.conv0 fwd bf16
pmc dgrad16s enc1.conv0 dgrad bf16
pmc wgrad16z dec5.conv1 wgrad bf16
pmc fwd32s enc1.conv0 fwd fp32
pmc dgrad32s enc1.conv0 dgrad fp32
pmc wgrad_strided enc1.conv0 wgrad fp32
pmc fwd_wino2 dec5.conv0 fwd fp32
echo collected
