#!/bin/bash
# round-4 baseline at HEAD: forward GEMM shapes by epilogue, kernel trace of the no-grad frame encoder, by-shape HBM traffic
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_base
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/scratch/fwd_shapes.py > $OUT/fwd_shapes.log 2>&1 || exit 1
rm -rf /tmp/prof_fwd
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_fwd -o t --output-format csv -- python3 $R/scratch/fwd_bench.py > $OUT/fwd_trace.log 2>&1 || exit 1
cp $(find /tmp/prof_fwd -name "*kernel_stats.csv" | head -1) $OUT/fwd_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcs_$c -o p --output-format csv -- python3 $R/scratch/gemm_bench.py 3 > $OUT/pmc_shape_$c.log 2>&1 || exit 1
done
python3 $R/scratch/pmc_by_shape.py $(find /tmp/pmcs_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmcs_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $OUT/r04_gemm_f16_hbm_traffic_by_shape_head.txt 2>&1
cat $OUT/fwd_shapes.log; tail -2 $OUT/fwd_trace.log; cat $OUT/r04_gemm_f16_hbm_traffic_by_shape_head.txt
