#!/bin/bash
mkdir -p gpurun_out/r03_x
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
timeout 600 python -m pytest tests/test_gpu_geom.py -x -q -k "alternate or placements" > gpurun_out/r03_x/pytest.log 2>&1; tail -3 gpurun_out/r03_x/pytest.log
UVOL_VARIANTS_WITH_ONLY=1 timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --only geo > gpurun_out/r03_x/bench_geo.json 2> gpurun_out/r03_x/bench_geo.err
grep "variant\|fault\|Fault" gpurun_out/r03_x/bench_geo.err | head; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_x/bench_geo.json').read().strip().splitlines()[-1]); print(d['value'], str(d.get('variants'))[:900])
except Exception as e: print('no json', e)
PY
