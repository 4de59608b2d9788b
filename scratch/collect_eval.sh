#!/bin/bash
# kernel stats of the eval leg only: bash scratch/collect_eval.sh r04
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r04}; OUT=$R/gpurun_out/profiles_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_eval
rocprofv3 --kernel-trace --stats -d /tmp/prof_eval -o t --output-format csv -- python3 $R/bench.py --mode eval --frames 24 --no-cpu-baseline > $OUT/${TAG}_bench_eval_f24_kernel_stats.log 2>&1
cp $(find /tmp/prof_eval -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_eval_f24_kernel_stats.csv
head -8 $OUT/${TAG}_bench_eval_f24_kernel_stats.csv | cut -c1-160
