#!/bin/bash
mkdir -p gpurun_out/r03_q
UVOL_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r03_q/bench.json 2> gpurun_out/r03_q/err.log
grep "uvol-timing" gpurun_out/r03_q/err.log | tail -6
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_q/bench.json').read().strip().splitlines()[-1])
print('fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1))
PY
