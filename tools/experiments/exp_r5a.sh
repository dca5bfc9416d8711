cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_a; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_geom.py -x -q -m gpu > $O/gpu_geom_tests.log 2>&1; echo "tests rc $?" >> $O/gpu_geom_tests.log
timeout 600 python bench.py --only geo --no-variants --no-cpu-baseline > $O/bench_geo.json 2> $O/bench_geo.err
timeout 900 python bench.py --no-variants --no-cpu-baseline > $O/bench.json 2> $O/bench.err
PMC_HALVES=geo PMC_TIMEOUT=200 bash tools/pmc_pack.sh r05_a
tail -3 $O/gpu_geom_tests.log; cat $O/bench_geo.json | head -c 1500; echo; cat $O/bench.json | head -c 600
