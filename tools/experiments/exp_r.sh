for f in 2160 2520 2880; do
  timeout 400 python bench.py --frames-per-step $f --steps 2 --warmup 1 --no-cpu-baseline 2>/tmp/err_$f.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('frames', $f, 'fps', round(d['value']), 'ms', round(d['ms_per_step']))
"
  tail -2 /tmp/err_$f.log | cut -c1-200
  rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -1
done
