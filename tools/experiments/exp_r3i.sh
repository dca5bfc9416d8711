#!/bin/bash
# lanes per wave of the lane-per-walker kernels on unrelated meshes (shuffled storage order)
mkdir -p gpurun_out/r03_i
for w in 1 2 4 8; do
UVOL_SIMT_W=$w timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --mesh-order shuffled > gpurun_out/r03_i/bench_w$w.json 2> gpurun_out/r03_i/err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_i/bench_w$w.json').read().strip().splitlines()[-1]); g=d['kernel_groups_ms_per_step']
print('W', $w, 'fps', round(d['value'],1), 'walk', round(g['geo.k4_eb_walk'],1), 'trav', round(g['geo.k5_traverse'],1), 'dedup', round(g['geo.k2_dedup'],1), 'entropy', round(g['geo.k7_entropy_encode'],1))
PY
done
