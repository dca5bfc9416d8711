for w in 1 2 4 8 16 32; do
  UVOL_ENTROPY_W=$w timeout 200 python bench.py --only geo --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('W', $w, 'entropy ms', round(d['kernel_groups_ms_per_step']['geo.k7_entropy_encode'],1), 'fps', round(d['value']))
"
done
