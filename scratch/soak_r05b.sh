#!/bin/bash
# longer sweeps of the round-5 code: 2 000 random shapes per kernel family, determinism soaks with more repeats and at the bench's batch shares
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/soak_r05b; mkdir -p $O
cd $R
timeout -k 10 900 python scratch/fuzz_kernels.py 2000 5051 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; grep "mismatches" $O/fuzz.log | tail -6
SOAK_B=32 SOAK_F=12 timeout -k 10 600 python scratch/soak_determinism.py 8 3 ft > $O/soak_ft32.log 2>&1; echo "soak ft b32 rc=$?"; tail -3 $O/soak_ft32.log
timeout -k 10 600 python scratch/soak_determinism.py 8 3 pt > $O/soak_pt.log 2>&1; echo "soak pt rc=$?"; tail -3 $O/soak_pt.log
timeout -k 10 300 python scratch/fuzz_f32_dma.py > $O/fuzz_f32.log 2>&1; echo "fuzz f32 rc=$?"; tail -2 $O/fuzz_f32.log
