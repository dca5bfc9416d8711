#!/bin/bash
# kernel statistics of the geometry half alone (no texture stream beside it)
mkdir -p gpurun_out/r03_s
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --only geo > $GRAFT_REPO_ROOT/gpurun_out/r03_s/bench_geo.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_s/err.log
cp $(find /tmp/prof_s -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_s/geometry_only_kernel_stats.csv
cd $GRAFT_REPO_ROOT; python - <<'PY'
import json,csv
d=json.loads(open('gpurun_out/r03_s/bench_geo.json').read().strip().splitlines()[-1])
print('geo only fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1))
print({k:round(v,1) for k,v in d['kernel_groups_ms_per_step'].items()})
rows=[]
for r in csv.DictReader(open('gpurun_out/r03_s/geometry_only_kernel_stats.csv')):
    rows.append((float(r['TotalDurationNs'])/1e6/int(r['Calls']), int(r['Calls']), r['Name'].split('(')[0]))
rows.sort(reverse=True)
for t,c,n in rows[:34]: print("%8.2f ms x %d %s"%(t,c,n))
PY
