#!/bin/bash
# host inputs (SURVEY 8(d) boundary): texture uploads in parts beside the kernels of the previous part
mkdir -p gpurun_out/r03_zy
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
timeout 600 python -m pytest tests/test_gpu_stress.py tests/test_gpu_tex.py -x -q > gpurun_out/r03_zy/pytest.log 2>&1; tail -2 gpurun_out/r03_zy/pytest.log
for p in 1 4 8; do
UVOL_TEX_PARTS=$p timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --host-inputs --frames-per-step 1080 > gpurun_out/r03_zy/bench_p$p.json 2> gpurun_out/r03_zy/err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_zy/bench_p$p.json').read().strip().splitlines()[-1])
print('parts', $p, 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1))
PY
done
