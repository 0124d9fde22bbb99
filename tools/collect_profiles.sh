#!/bin/bash
# Round profile collection (run through gpurun from the repo root): kernel statistics of the bench steps, the isolated
# roofline launches, and the PMC passes of the kernels VERDICT.md names.  Output under gpurun_out/profiles_$R/ ; copy
# the summaries into profiles/ (tools/collect_profiles.sh does not touch profiles/ itself).
set -e
export TMPDIR=/tmp PYTHONPATH=$PWD
ROOT=$PWD
RR=${ROUND:-r02}
OUT=$ROOT/gpurun_out/profiles_$RR
mkdir -p $OUT
cd /tmp
stats() {  # tag, bench args...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/$tag.log 2>&1 || echo "$tag failed"
  cp $(ls $OUT/$tag/*/*kernel_stats.csv | head -1) $OUT/${RR}_${tag}_kernel_stats.csv
  tail -1 $OUT/$tag.log | cut -c1-400
}
stats bench
stats bench_bf16 --precision bf16
stats bench_cfg3 --config cfg3
stats bench_cfg4 --config cfg4
stats bench_cfg5 --config cfg5
# the timed roofline layers alone (three repetitions of each isolated launch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roof_fp32 -- python3 $ROOT/bench.py --roofline-only > $OUT/roof_fp32.log 2>&1
cp $(ls $OUT/roof_fp32/*/*kernel_stats.csv | head -1) $OUT/${RR}_roofline_fp32_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roof_bf16 -- python3 $ROOT/bench.py --roofline-only --precision bf16 > $OUT/roof_bf16.log 2>&1
cp $(ls $OUT/roof_bf16/*/*kernel_stats.csv | head -1) $OUT/${RR}_roofline_bf16_kernel_stats.csv
tail -1 $OUT/roof_fp32.log | cut -c1-600; tail -1 $OUT/roof_bf16.log | cut -c1-600
cd $ROOT
LAYER=dec5.conv1 WHAT=fwd DTYPE=bf16 ITERS=10 TAG=${RR}_fwd16z tools/pmc_kernel.sh > $OUT/pmc_fwd16z.txt 2>&1 || true
LAYER=dec5.conv1 WHAT=wgrad DTYPE=bf16 ITERS=10 TAG=${RR}_wgrad16 tools/pmc_kernel.sh > $OUT/pmc_wgrad16.txt 2>&1 || true
LAYER=dec5.conv0 WHAT=fwd DTYPE=fp32 ITERS=5 TAG=${RR}_fwd_wino2 tools/pmc_kernel.sh > $OUT/pmc_fwd_wino2.txt 2>&1 || true
cp gpurun_out/pmc_${RR}_fwd16z/summary.txt $OUT/${RR}_pmc_fwd16z_summary.txt || true
cp gpurun_out/pmc_${RR}_wgrad16/summary.txt $OUT/${RR}_pmc_wgrad16_summary.txt || true
cp gpurun_out/pmc_${RR}_fwd_wino2/summary.txt $OUT/${RR}_pmc_fwd_wino2_summary.txt || true
echo collected
