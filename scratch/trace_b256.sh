cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_o
rocprofv3 --kernel-trace --stats -d /tmp/prof_o -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 --roofline-steps 0 --vit-forward-iters 0 > $R/gpurun_out/tr256.log 2>&1
python3 $R/scratch/trace_gaps.py $(find /tmp/prof_o -name "*kernel_trace.csv" | head -1) 12 > $R/gpurun_out/tr256_breakdown.txt 2>&1
cp $(find /tmp/prof_o -name "*kernel_trace.csv" | head -1) $R/gpurun_out/tr256_trace.csv
