# checkpoint of the build: all GPU tests, the default bench line, geometry alone, the boundary variants, HBM traffic of the geometry kernels
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_i; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc $?" >> $O/gpu_tests.log
timeout 600 python bench.py --only geo --no-variants --no-cpu-baseline > $O/bench_geo.json 2> $O/bench_geo.err
timeout 900 python bench.py --no-variants --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 900 python bench.py --no-variants --no-cpu-baseline --host-inputs > $O/bench_host.json 2> $O/bench_host.err
timeout 900 python bench.py --no-variants --no-cpu-baseline --host-inputs --host-pinned > $O/bench_host_pinned.json 2> $O/bench_host_pinned.err
PMC_HALVES=geo PMC_TIMEOUT=200 bash tools/pmc_pack.sh r05_i
tail -3 $O/gpu_tests.log
python - <<PY
import json
for t in ("bench_geo","bench","bench_host","bench_host_pinned"):
    try:
        d=json.load(open("$O/%s.json"%t)); print(t, round(d["value"]), "fps", round(d["ms_per_step"]), "ms")
    except Exception as e: print(t, "FAILED", e)
PY
