for c in 256 512 768; do
  for m in "--only tex" ""; do
  UVOL_SEL_LCAP=$c timeout 300 python bench.py $m --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); g=d['kernel_groups_ms_per_step']; print('lcap', $c, '$m', 'fps', round(d['value']), 'ms', round(d['ms_per_step']), 'selcb', round(g.get('tex.k10_selector_codebook',0)))
"
  done
done
