cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_pt
rocprofv3 --kernel-trace --stats -d /tmp/prof_pt -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --mode pretrain --steps 4 --warmup 2 --roofline-steps 0 --single-stream --vit-forward-iters 0 > $R/gpurun_out/pt_trace.log 2>&1
python3 $R/scratch/trace_gaps.py $(find /tmp/prof_pt -name "*kernel_trace.csv" | head -1) 70 > $R/gpurun_out/pt_breakdown.txt 2>&1
