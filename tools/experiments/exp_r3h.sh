#!/bin/bash
# kernel statistics of the texture half alone + parity of the texture half; 64-bit against 32-bit statistics words
mkdir -p gpurun_out/r03_h
timeout 900 python -m pytest tests/test_gpu_tex.py -x -q > gpurun_out/r03_h/pytest_tex.log 2>&1; tail -2 gpurun_out/r03_h/pytest_tex.log
for w in 64 32; do
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_h
UVOL_SEL_WORD=$w timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o tex -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --only tex > $GRAFT_REPO_ROOT/gpurun_out/r03_h/bench_tex_$w.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_h/err.log
cp $(find /tmp/prof_h -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_h/tex_kernel_stats_$w.csv
echo "== word $w"; grep -E "k_sel_s|k_sel_split" $GRAFT_REPO_ROOT/gpurun_out/r03_h/tex_kernel_stats_$w.csv | cut -c1-170
cd $GRAFT_REPO_ROOT; python - <<PY
import json
d=json.loads(open('gpurun_out/r03_h/bench_tex_$w.json').read().strip().splitlines()[-1])
print('tex only fps', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'selcb', round(d['kernel_groups_ms_per_step']['tex.k10_selector_codebook'],1))
PY
done
