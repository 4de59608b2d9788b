#!/bin/bash
# forward-only checks: fold tests, forward bench, kernel trace of the forward
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_fwd
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_fold.py -x -q -s 2>&1 | tail -12 || exit 1
timeout -k 10 300 python scratch/fwd_bench.py || exit 1
HMMC_FOLD_LN=0 timeout -k 10 300 python scratch/fwd_bench.py || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fwd
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_fwd -o t --output-format csv -- python3 $R/scratch/fwd_bench.py > $OUT/fwd_trace.log 2>&1 || exit 1
cp $(find /tmp/prof_fwd -name "*kernel_stats.csv" | head -1) $OUT/fwd_kernel_stats.csv
head -14 $OUT/fwd_kernel_stats.csv | cut -c1-200
