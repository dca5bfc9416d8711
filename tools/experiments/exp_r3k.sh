#!/bin/bash
# decode path after the chunked value recurrences: parity + timing + kernel statistics
mkdir -p gpurun_out/r03_k
timeout 900 python -m pytest tests/test_gpu_geom.py -x -q -k "decode or decoder or roundtrip" > gpurun_out/r03_k/pytest.log 2>&1; tail -2 gpurun_out/r03_k/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o gdec -- python $GRAFT_REPO_ROOT/tools/gdec_timing.py 1920 > $GRAFT_REPO_ROOT/gpurun_out/r03_k/gdec_1920_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_k/err.log
cp $(find /tmp/prof_k -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_k/gdec_kernel_stats.csv
grep -E "gdec_pred|gdec_conn|gdec_rans|traverse" $GRAFT_REPO_ROOT/gpurun_out/r03_k/gdec_kernel_stats.csv | cut -c1-140
cat $GRAFT_REPO_ROOT/gpurun_out/r03_k/gdec_1920_prof.json
