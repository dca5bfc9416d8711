for c in "" "4:7" "8:7f" "8:3f" "8:1f" "16:7fff" "16:3fff" "16:fff" "3:3" "5:f" "6:1f"; do
  a=""; [ -n "$c" ] && a="--tex-cu $c"
  timeout 300 python bench.py $a --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); g=d['kernel_groups_ms_per_step']; print('tex-cu [$c]', 'fps', round(d['value']), 'ms', round(d['ms_per_step']), 'walk', round(g['geo.k4_eb_walk']), 'trav', round(g['geo.k5_traverse']), 'selcb', round(g['tex.k10_selector_codebook']), 'fit', round(g['tex.k9_endpoint_fit']))
"
done
