for a in "--geo-streams 1" "--geo-streams 2" "--geo-streams 3" "--geo-streams 2 --only geo" "--geo-streams 3 --only geo"; do
  timeout 400 python bench.py $a --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$a', 'fps', round(d['value']), 'ms', round(d['ms_per_step']))
"
done
