for c in 0 243 241 247; do
  timeout 300 python bench.py --cu-split $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); g=d['kernel_groups_ms_per_step']; print('cu-split', $c, 'fps', round(d['value']), 'ms', round(d['ms_per_step']), 'trav', round(g['geo.k5_traverse']), 'selcb', round(g['tex.k10_selector_codebook']))
"
done
