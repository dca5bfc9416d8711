python -m pytest tests/test_gpu_geom.py -x -q 2>&1 | tail -2
for f in 2160 2400 2520; do
  for m in "--only geo" ""; do
  timeout 400 python bench.py $m --frames-per-step $f --steps 2 --warmup 1 --no-cpu-baseline 2>/tmp/err.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('frames', $f, '[$m]', 'fps', round(d['value']), 'ms', round(d['ms_per_step']), 'ws', d['config']['geometry_workspace_bytes_per_frame'])
"
  tail -1 /tmp/err.log | cut -c1-160
  done
done
