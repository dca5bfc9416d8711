#!/bin/bash
# frames in flight per step against device memory, on the final build: 2560 (16.6 GB below the size that no longer fits) with all variants
mkdir -p gpurun_out/r03_zw
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
for n in 2560 2560; do
timeout 400 python bench.py --no-cpu-baseline --frames-per-step $n > gpurun_out/r03_zw/bench_$n.json 2> gpurun_out/r03_zw/err.log
if [ ! -s gpurun_out/r03_zw/bench_$n.json ]; then echo "N $n FAILED"; tail -2 gpurun_out/r03_zw/err.log; continue; fi
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_zw/bench_$n.json').read().strip().splitlines()[-1])
print('N', $n, 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), {k:round(v.get('frames_per_s', v.get('value',0)),1) if isinstance(v,dict) else v for k,v in d['variants'].items()})
PY
done
