#!/bin/bash
mkdir -p gpurun_out/r03_t
timeout 900 python -m pytest tests/test_gpu_geom.py -x -q -k "bit_exact or edge or soups or storage or 256" > gpurun_out/r03_t/pytest.log 2>&1; tail -2 gpurun_out/r03_t/pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r03_t/bench_$i.json 2> gpurun_out/r03_t/err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_t/bench_$i.json').read().strip().splitlines()[-1])
print('fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_groups_ms_per_step'].items() if k.startswith('geo')})
PY
done
