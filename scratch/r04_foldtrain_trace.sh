#!/bin/bash
# kernel stats of the default step with and without the training fold (single stream), same box
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_foldtrain
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 vit; do
  rm -rf /tmp/prof_$v
  HMMC_FOLD_LN_TRAIN=$v timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 --roofline-steps 2 --single-stream --vit-forward-iters 0 > $OUT/bench_$v.log 2>&1 || exit 1
  cp $(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1) $OUT/stats_$v.csv
  tail -1 $OUT/bench_$v.log | cut -c1-200
done
