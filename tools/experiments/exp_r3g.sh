#!/bin/bash
# round 3: incremental selector statistics - parity tests of the texture half + the default bench line
mkdir -p gpurun_out/r03_g
timeout 900 python -m pytest tests/test_gpu_tex.py -x -q > gpurun_out/r03_g/pytest_tex.log 2>&1; tail -3 gpurun_out/r03_g/pytest_tex.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r03_g/bench.json 2> gpurun_out/r03_g/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_g/bench.json').read().strip().splitlines()[-1])
print('fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1))
print({k:round(v,1) for k,v in d['kernel_groups_ms_per_step'].items()})
PY
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --only tex > gpurun_out/r03_g/bench_tex.json 2>> gpurun_out/r03_g/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_g/bench_tex.json').read().strip().splitlines()[-1])
print('tex only fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1))
print({k:round(v,1) for k,v in d['kernel_groups_ms_per_step'].items()})
PY
