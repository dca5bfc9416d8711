#!/bin/bash
# kernel statistics of the default step after the vectorised flag kernels
mkdir -p gpurun_out/r03_o
timeout 900 python -m pytest tests/test_gpu_geom.py -x -q -k "bit_exact or edge or soups or storage" > gpurun_out/r03_o/pytest.log 2>&1; tail -2 gpurun_out/r03_o/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_o -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $GRAFT_REPO_ROOT/gpurun_out/r03_o/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_o/err.log
cp $(find /tmp/prof_o -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_o/kernel_stats.csv
cd $GRAFT_REPO_ROOT; python - <<'PY'
import json,csv
d=json.loads(open('gpurun_out/r03_o/bench.json').read().strip().splitlines()[-1])
print('fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1))
print({k:round(v,1) for k,v in d['kernel_groups_ms_per_step'].items() if k.startswith('geo')})
for r in csv.DictReader(open('gpurun_out/r03_o/kernel_stats.csv')):
    n=r['Name'].split('(')[0]
    if n in ('k_seam_bits','k_eb_event_compact','k_valence_init','k_renumber_seams','k_aseg_a','k_aseg_b','k_eb_valence','k_dd_resolve','k_edge_match','k_eb_event_flags'): print(n, r['Calls'], round(float(r['AverageNs'])/1e6,2))
PY
