for w in 8 16 32; do
  for m in "--only geo" ""; do
  UVOL_SIMT_W=$w timeout 300 python bench.py $m --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); g=d['kernel_groups_ms_per_step']; print('W', $w, '[$m]', 'fps', round(d['value']), 'walk', round(g['geo.k4_eb_walk']), 'trav', round(g['geo.k5_traverse']))
"
  done
done
