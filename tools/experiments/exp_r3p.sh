#!/bin/bash
mkdir -p gpurun_out/r03_p
timeout 900 python -m pytest tests/test_gpu_geom.py -x -q -k "bit_exact or edge or soups or storage or dedup" > gpurun_out/r03_p/pytest.log 2>&1; tail -2 gpurun_out/r03_p/pytest.log
for a in "" "--mesh-order shuffled"; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants $a > gpurun_out/r03_p/bench.json 2> gpurun_out/r03_p/err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_p/bench.json').read().strip().splitlines()[-1])
print('[$a] fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_groups_ms_per_step'].items() if k.startswith('geo')})
PY
done
