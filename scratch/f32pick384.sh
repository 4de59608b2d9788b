#!/bin/bash
# every fp32 GEMM kernel on the temporal transformer's shapes at 384 / 1536 / 3072 tokens (HMMC_F32_PICK: scratch build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 0 1 2 3 4 7 9; do
  echo "== pick $k"
  HMMC_F32_PICK=$k HMMC_LIB=$R/scratch/_dbg/libhmmc_f32pick.so timeout -k 10 120 python scratch/gemm32_pick.py 2>/dev/null | grep "T384\|T1536"
done
