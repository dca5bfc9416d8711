#!/bin/bash
# 300..500-frame jobs: traversers with the vertex bitmap in L2 (6 per CU) against the lane form
mkdir -p gpurun_out/r03_w
for n in 300 450 510; do for v in 0 1; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --frames-per-step $n --traverse-vbits-l2 $v > gpurun_out/r03_w/bench.json 2> gpurun_out/r03_w/err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_w/bench.json').read().strip().splitlines()[-1]); g=d['kernel_groups_ms_per_step']
print('N', $n, 'vbits_l2', $v, 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'trav', round(g['geo.k5_traverse'],1), 'walk', round(g['geo.k4_eb_walk'],1))
PY
done; done
