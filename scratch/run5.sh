#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/run5; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > $O/attn.log 2>&1; echo "attn tests rc=$?"; tail -5 $O/attn.log
for v in 0 1 1 0; do
  echo "== no_attn_pipe=$v"; if [ $v = 1 ]; then export HMMC_NO_ATTN_PIPE=1; else unset HMMC_NO_ATTN_PIPE; fi
  timeout -k 10 120 python scratch/attn_bench.py 2>/dev/null | grep bwd
done
