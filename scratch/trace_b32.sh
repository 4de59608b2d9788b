#!/bin/bash
# b = 32 overlapped kernel trace (timestamps kept) + un-profiled A/A timings: bash scratch/trace_b32.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-t}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
  timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-hbm-roofline --batch 32 --steps 30 --warmup 5 --roofline-steps 0 --vit-forward-iters 0 --reserve-cus 16 > $OUT/plain$i.log 2>&1 || exit 1
  python3 -c "import json,sys; d=json.loads(open('$OUT/plain$i.log').read().strip().splitlines()[-1]); print('b32 plain', d['ms_per_step'])"
done
rm -rf /tmp/prof_t
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-hbm-roofline --batch 32 --steps 6 --warmup 2 --roofline-steps 0 --vit-forward-iters 0 --reserve-cus 16 > $OUT/prof.log 2>&1 || exit 1
cp $(find /tmp/prof_t -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
cp $(find /tmp/prof_t -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 $R/scratch/trace_gaps.py $OUT/kernel_trace.csv 30 > $OUT/breakdown.txt 2>&1
head -3 $OUT/breakdown.txt
