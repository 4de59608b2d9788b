#!/bin/bash
# b = 32 (the 8-GPU share of config 2) kernel stats + per-step breakdowns, single-stream and overlapped: bash scratch/collect_b32.sh r04
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in "" o; do
  if [ "$m" = "" ]; then SS=--single-stream; else SS=; fi
  rm -rf /tmp/prof_b32$m
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b32$m -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-hbm-roofline --batch 32 --steps 6 --warmup 2 --roofline-steps 0 $SS --vit-forward-iters 0 > $OUT/${TAG}_b32$m.log 2>&1 || exit 1
done
cp $(find /tmp/prof_b32 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_b32_kernel_stats_single_stream.csv
python3 $R/scratch/trace_gaps.py $(find /tmp/prof_b32 -name "*kernel_trace.csv" | head -1) 45 > $OUT/${TAG}_bench_b32_kernel_breakdown_single_stream.txt 2>&1
python3 $R/scratch/trace_gaps.py $(find /tmp/prof_b32o -name "*kernel_trace.csv" | head -1) 45 > $OUT/${TAG}_bench_b32_kernel_breakdown.txt 2>&1
head -3 $OUT/${TAG}_bench_b32_kernel_breakdown.txt; head -3 $OUT/${TAG}_bench_b32_kernel_breakdown_single_stream.txt
