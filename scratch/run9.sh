#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/run9; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -3 $O/suite.log
[ $rc = 0 ] || exit 1
bash scratch/ab_tree.sh b32 --batch 32 --steps 40 --warmup 5 --reserve-cus 16
