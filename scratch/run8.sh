#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/run8; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_pretrain.py tests/test_gpu_fp32_regime.py tests/test_gpu_ddp.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -3 $O/suite.log
[ $rc = 0 ] || exit 1
bash scratch/ab_tree.sh pt --mode pretrain --steps 10 --warmup 3
