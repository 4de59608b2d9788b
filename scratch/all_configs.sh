for args in "" "--mode pretrain" "--batch 128" "--batch 64" "--batch 32" "--clip ViT-B/16 --frames 24 --batch 16"; do
  echo "== $args"
  python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; v=d.get('vit_forward') or {}
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'avg_us', r['avg_launch_us'], {k:(round(x['tflops']),x['launches']) for k,x in r['by_layout'].items()}, 'vit', v.get('ms'), v.get('frac_of_mfma_peak'), 'mem', d.get('peak_device_memory_gib'))
"
done
