#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/run2; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ddp.py -x -q -s -m gpu -k "two_ranks_equal_single or rccl_one_rank" > $O/ddp.log 2>&1; echo "ddp rc=$?"; grep -a "2 ranks vs" $O/ddp.log; tail -3 $O/ddp.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; echo "suite rc=$?"; tail -5 $O/suite.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > $O/b256.log 2>&1; echo "b256 rc=$?"; tail -c 600 $O/b256.log
timeout -k 10 600 python bench.py --no-cpu-baseline --regime fp32 --steps 3 --warmup 1 --roofline-steps 1 --no-hbm-roofline > $O/fp32.log 2>&1; echo "fp32 rc=$?"; tail -c 3000 $O/fp32.log
