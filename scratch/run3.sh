#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/run3; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -m gpu > $O/full.log 2>&1; echo "full rc=$?"; grep -a "config\|passed\|failed\|Error\|assert" $O/full.log | head -40
timeout -k 10 600 python -m pytest tests/test_gpu_fold.py -q -s -m gpu -k "large_variance or unsupported" > $O/fold.log 2>&1; echo "fold rc=$?"; grep -a "outlier\|layer\|dx \|passed\|failed\|Error" $O/fold.log | head -80
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -q -m gpu -k "stale or optimizer_takes" > $O/opt.log 2>&1; echo "opt rc=$?"; tail -3 $O/opt.log
