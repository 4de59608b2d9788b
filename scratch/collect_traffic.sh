#!/bin/bash
# Only the HBM-traffic record of gemm_f16_kernel (the two --pmc passes of collect_profiles.sh): bash scratch/collect_traffic.sh r04
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="--steps 2 --warmup 1 --no-cpu-baseline --no-hbm-roofline --roofline-steps 0 --vit-forward-iters 0 --unfolded-steps 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python3 $R/bench.py $CMD > $OUT/pmc_$c.log 2>&1
done
python3 $R/scratch/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) \
  $OUT/${TAG}_gemm_f16_hbm_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py $CMD" 3
