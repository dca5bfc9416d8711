#!/bin/bash
# stream priorities of the two contexts now that the geometry context is the longer one (repeat of the two best settings)
mkdir -p gpurun_out/r03_u
for c in "1 0" "0 1" "1 0" "0 1" "1 0" "0 1"; do
set -- $c
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-variants --tex-priority $1 --geo-priority $2 > gpurun_out/r03_u/bench_$1$2.json 2> gpurun_out/r03_u/err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_u/bench_$1$2.json').read().strip().splitlines()[-1]); g=d['kernel_groups_ms_per_step']
print('tex-prio $1 geo-prio $2 fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1))
PY
done
