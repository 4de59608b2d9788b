#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/run7; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -3 $O/suite.log
[ $rc = 0 ] || exit 1
bash scratch/ab_tree.sh b32 --batch 32 --steps 40 --warmup 5 --reserve-cus 16 && bash scratch/ab_tree.sh b256 --steps 10 --warmup 3
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > $O/b256.log 2>&1; python3 - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/run7/b256.log') if l.startswith('{')][-1])
print('default', d['ms_per_step'], 'unfolded', d['ms_per_step_unfolded'], 'roof', d['roofline']['frac'], 'vit', d['vit_forward']['ms'], d['vit_forward']['frac_of_mfma_peak'])
for h in d['roofline_hbm']: print('  ', h['kernel'][:50], h['us'], h['frac_of_hbm_peak'])
PY
