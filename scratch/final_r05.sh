#!/bin/bash
# final numbers of the round: suite, default bench line (with cpu baseline as the driver runs it), b = 32 share, pre-train, ViT-B/16 share, fp32 regime, eval
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final_r05; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -3 $O/suite.log
[ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/default.log 2>&1; echo "default rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --batch 32 --reserve-cus 16 --steps 40 --warmup 5 > $O/b32.log 2>&1; echo "b32 rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --batch 32 --steps 40 --warmup 5 > $O/b32_noreserve.log 2>&1; echo "b32nr rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --mode pretrain --steps 10 --warmup 3 > $O/pretrain.log 2>&1; echo "pt rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --clip ViT-B/16 --frames 24 --batch 16 --steps 10 --warmup 3 > $O/vitb16.log 2>&1; echo "b16 rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --mode eval --frames 24 > $O/eval.log 2>&1; echo "eval rc=$?"
for f in default b32 b32_noreserve pretrain vitb16; do python3 - $O/$f.log <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
v=d.get('vit_forward') or {}
print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], 'unf', d.get('ms_per_step_unfolded'), 'roof', d['roofline']['frac'], 'vit', v.get('ms'), v.get('frac_of_mfma_peak'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'mem', d['peak_device_memory_gib'])
PY
done
tail -c 400 $O/eval.log
