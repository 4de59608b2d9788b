#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r04}; OUT=$R/gpurun_out/profiles_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/hb_$c; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/hb_$c -o p --output-format csv -- python3 $R/scratch/hbm_kernels.py > $OUT/hbm_$c.log 2>&1 || exit 1; done
rm -rf /tmp/hb_stats; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/hb_stats -o t --output-format csv -- python3 $R/scratch/hbm_kernels.py > $OUT/hbm_stats.log 2>&1 || exit 1
python3 $R/scratch/hbm_kernels_summary.py $(find /tmp/hb_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/hb_WRITE_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/hb_stats -name "*kernel_stats.csv" | head -1) > $OUT/${TAG}_hbm_kernels_traffic.txt 2>&1
cat $OUT/${TAG}_hbm_kernels_traffic.txt
