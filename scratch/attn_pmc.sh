#!/bin/bash
# SQ counters of the short attention kernels (scratch/attn_bench.py): bash scratch/attn_pmc.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-t}
OUT=$R/gpurun_out/attn_pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/asq$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/asq$i -o p --output-format csv -- python3 $R/scratch/attn_bench.py > $OUT/run$i.log 2>&1 || echo "pass $i failed"
done
PMC_KERNELS=attn_ python3 $R/scratch/pmc_summary.py $(find /tmp/asq1 /tmp/asq2 /tmp/asq3 /tmp/asq4 /tmp/asq5 -name "*counter_collection.csv") > $OUT/attn_sq_counters.txt 2>&1
cat $OUT/attn_sq_counters.txt
