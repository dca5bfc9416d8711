O=gpurun_out/r02_x; mkdir -p $O
for t in 1 7 8 9 10 11 12 13 14; do
  UVOL_EXP_ENT=$t timeout 200 python bench.py --only geo --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('stream', $t, 'entropy group ms', round(d['kernel_groups_ms_per_step']['geo.k7_entropy_encode'],1))
"
done
