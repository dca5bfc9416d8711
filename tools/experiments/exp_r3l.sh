#!/bin/bash
# mid-size jobs with the wave-per-walker traversers whose face flags live in the record blocks (short timeouts: a hang must not cost minutes)
mkdir -p gpurun_out/r03_l
timeout 600 python -m pytest tests/test_gpu_geom.py -x -q -k "placements" > gpurun_out/r03_l/pytest.log 2>&1; tail -2 gpurun_out/r03_l/pytest.log
for n in 300 450 600 750; do
timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --frames-per-step $n > gpurun_out/r03_l/bench_n$n.json 2> gpurun_out/r03_l/err_$n.log
if [ ! -s gpurun_out/r03_l/bench_n$n.json ]; then echo "N $n FAILED"; tail -3 gpurun_out/r03_l/err_$n.log; exit 1; fi
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_l/bench_n$n.json').read().strip().splitlines()[-1]); g=d['kernel_groups_ms_per_step']
print('N', $n, 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), {k:round(v,1) for k,v in g.items() if k.startswith('geo') and v > 8})
PY
done
