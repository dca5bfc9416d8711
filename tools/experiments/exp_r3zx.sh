#!/bin/bash
# per-pass barrier + delayed start of the texture contexts (their heavy first kernels beside the geometry's walks instead of its front end)
mkdir -p gpurun_out/r03_zx
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
for a in "" "--lockstep" "--lockstep --tex-delay-ms 100" "--lockstep --tex-delay-ms 150" "--lockstep --tex-delay-ms 200" "--lockstep --tex-delay-ms 260"; do
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-variants $a > gpurun_out/r03_zx/bench.json 2> gpurun_out/r03_zx/err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_zx/bench.json').read().strip().splitlines()[-1]); g=d['kernel_groups_ms_per_step']
print('[$a] fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'dedup', round(g['geo.k2_dedup'],1), 'walk', round(g['geo.k4_eb_walk'],1), 'trav', round(g['geo.k5_traverse'],1), 'fit', round(g['tex.k9_endpoint_fit'],1))
PY
done
