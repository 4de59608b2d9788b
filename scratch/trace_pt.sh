#!/bin/bash
# pre-training (config 4) and b = 32 kernel traces with timestamps: bash scratch/trace_pt.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-t}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pt -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-hbm-roofline --mode pretrain --steps 4 --warmup 2 --roofline-steps 0 --vit-forward-iters 0 --unfolded-steps 0 > $OUT/pt.log 2>&1 || exit 1
cp $(find /tmp/prof_pt -name "*kernel_trace.csv" | head -1) $OUT/pt_kernel_trace.csv
python3 $R/scratch/trace_gaps.py $OUT/pt_kernel_trace.csv 40 > $OUT/pt_breakdown.txt 2>&1
head -2 $OUT/pt_breakdown.txt
rm -rf /tmp/prof_b32
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b32 -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-hbm-roofline --batch 32 --steps 6 --warmup 2 --roofline-steps 0 --vit-forward-iters 0 --unfolded-steps 0 --reserve-cus 16 > $OUT/b32.log 2>&1 || exit 1
cp $(find /tmp/prof_b32 -name "*kernel_trace.csv" | head -1) $OUT/b32_kernel_trace.csv
python3 $R/scratch/trace_gaps.py $OUT/b32_kernel_trace.csv 40 > $OUT/b32_breakdown.txt 2>&1
head -2 $OUT/b32_breakdown.txt
