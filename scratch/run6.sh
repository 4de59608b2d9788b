#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/run6; mkdir -p $O
cd $R
bash scratch/attn_abl.sh
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > $O/attn.log 2>&1; echo "attn tests rc=$?"; tail -2 $O/attn.log
