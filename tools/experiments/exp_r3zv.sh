#!/bin/bash
# measurement of the new default (2560 frames per step): bench x2, rocprof kernel statistics, GPU suite
TAG=r03_y2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench_run2.json 2>> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $O/bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/kt -name bench_kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/kt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python - <<'PY'
import json
for f in ('bench','bench_run2','bench_under_rocprof'):
    d=json.loads(open('gpurun_out/r03_y2/%s.json'%f).read().strip().splitlines()[-1]); print(f, round(d['value'],1), round(d['ms_per_step'],1), d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
