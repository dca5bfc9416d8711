#!/bin/bash
# full GPU suite + smoke + default bench line
mkdir -p gpurun_out/r03_m
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03_m/pytest_gpu.log 2>&1; tail -4 gpurun_out/r03_m/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r03_m/bench.json 2> gpurun_out/r03_m/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_m/bench.json').read().strip().splitlines()[-1])
print('fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'roofline', d['roofline']['frac'])
print({k:round(v,1) for k,v in d['kernel_groups_ms_per_step'].items()})
print({k:(round(v.get('frames_per_s',v.get('value',0)),1)) for k,v in d['variants'].items()})
PY
