#!/bin/bash
# geometry streams: does a second geometry context fill the chip now that the texture context is shorter?
mkdir -p gpurun_out/r03_n
for g in 1 2 3; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --geo-streams $g > gpurun_out/r03_n/bench_g$g.json 2> gpurun_out/r03_n/err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_n/bench_g$g.json').read().strip().splitlines()[-1]); g=d['kernel_groups_ms_per_step']
print('geo-streams', $g, 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'trav', round(g['geo.k5_traverse'],1), 'walk', round(g['geo.k4_eb_walk'],1))
PY
done
