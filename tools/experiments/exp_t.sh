for a in "" "--lockstep" "--lockstep --tex-delay-ms 60" "--lockstep --tex-delay-ms 100" "--lockstep --tex-delay-ms 150" "--lockstep --tex-delay-ms 250" "--tex-delay-ms 100"; do
  timeout 400 python bench.py $a --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$a]', 'fps', round(d['value']), 'ms', round(d['ms_per_step']))
"
done
