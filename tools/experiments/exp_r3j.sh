#!/bin/bash
# lattice vs shuffled, lanes per wave 1 vs auto, geometry alone
mkdir -p gpurun_out/r03_j
run() { # name, env..., args
name=$1; shift
env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --only geo $ARGS > gpurun_out/r03_j/bench_$name.json 2> gpurun_out/r03_j/err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_j/bench_$name.json').read().strip().splitlines()[-1]); g=d['kernel_groups_ms_per_step']
print('$name', 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'walk', round(g['geo.k4_eb_walk'],1), 'trav', round(g['geo.k5_traverse'],1), 'dedup', round(g['geo.k2_dedup'],1), 'corner', round(g['geo.k3_corner_table'],1), 'renum', round(g['geo.k4b_renumber_seams'],1))
PY
}
ARGS="" run lattice_auto A=1
ARGS="" run lattice_w1 UVOL_SIMT_W=1
ARGS="--mesh-order shuffled" run shuffled_auto A=1
ARGS="--mesh-order shuffled" run shuffled_w16 UVOL_SIMT_W=16
