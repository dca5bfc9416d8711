#!/bin/bash
# decode timing of the final build (after the connectivity-machine change) + the default bench line once more
mkdir -p gpurun_out/r03_zz
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
python tools/gdec_timing.py 1920 > gpurun_out/r03_zz/gdec_timing.json 2> gpurun_out/r03_zz/err.log; cat gpurun_out/r03_zz/gdec_timing.json
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03_zz/bench.json 2>> gpurun_out/r03_zz/err.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_zz/bench.json').read().strip().splitlines()[-1])
print('fps', round(d['value'],1), {k:round(v.get('frames_per_s', v.get('value',0)),1) for k,v in d['variants'].items()})
PY
