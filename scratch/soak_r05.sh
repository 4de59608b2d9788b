#!/bin/bash
# round-5 sweeps outside the test suite: random-shape fuzz of the kernels, determinism soak (fine-tune and pre-train, every stream mode), poison soak
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/soak_r05; mkdir -p $O
cd $R
timeout -k 10 420 python scratch/fuzz_kernels.py 400 2028 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -4 $O/fuzz.log
timeout -k 10 300 python scratch/soak_determinism.py 6 3 ft > $O/soak_ft.log 2>&1; echo "soak ft rc=$?"; tail -3 $O/soak_ft.log
timeout -k 10 300 python scratch/soak_determinism.py 4 3 pt > $O/soak_pt.log 2>&1; echo "soak pt rc=$?"; tail -3 $O/soak_pt.log
timeout -k 10 300 python scratch/soak_poison.py > $O/poison.log 2>&1; echo "poison rc=$?"; tail -4 $O/poison.log
