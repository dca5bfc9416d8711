#!/bin/bash
# one traverser per wave + a helper wave per four (k_traverse_simt_pf) on unrelated meshes (shuffled storage order); short timeouts
mkdir -p gpurun_out/r03_y
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
for pf in 1 0; do for a in "--only geo"; do
UVOL_SIMT_PF=$pf timeout 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --mesh-order shuffled $a > gpurun_out/r03_y/bench.json 2> gpurun_out/r03_y/err.log
if [ ! -s gpurun_out/r03_y/bench.json ]; then echo "pf $pf [$a] FAILED"; tail -3 gpurun_out/r03_y/err.log; exit 1; fi
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_y/bench.json').read().strip().splitlines()[-1]); g=d['kernel_groups_ms_per_step']
print('pf', $pf, '[$a]', 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'trav', round(g['geo.k5_traverse'],1), 'walk', round(g['geo.k4_eb_walk'],1))
PY
done; done
