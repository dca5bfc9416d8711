#!/bin/bash
# small jobs after the late join of the auxiliary stream
mkdir -p gpurun_out/r03_l
timeout 900 python -m pytest tests/test_gpu_geom.py -x -q -k "placements or bit_exact or 256 or lane_per" > gpurun_out/r03_l/pytest.log 2>&1; tail -2 gpurun_out/r03_l/pytest.log
for n in 150 300 600 1200; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --frames-per-step $n > gpurun_out/r03_l/bench_n$n.json 2> gpurun_out/r03_l/err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_l/bench_n$n.json').read().strip().splitlines()[-1]); g=d['kernel_groups_ms_per_step']
print('N', $n, 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), {k:round(v,1) for k,v in g.items() if k.startswith('geo') and v > 5})
PY
done
