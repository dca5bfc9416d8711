# round 4: lanes (groups of a call on their own streams, front ends chained) against one group per call; enqueued passes against blocking calls
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_geom.py -x -q -k "small_batch or 1280 or 256_full or alternate or enqueue or overflow" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="python bench.py --only geo --no-variants --no-cpu-baseline --parity-frames 0 --steps 4"
for L in 1 2 4 8; do
  UVOL_GEO_LANES=$L timeout 300 $B > $O/geo_lanes${L}.json 2>> $O/sweep.err
  UVOL_GEO_LANES=$L timeout 300 $B --blocking-calls > $O/geo_lanes${L}_blocking.json 2>> $O/sweep.err
done
UVOL_GEO_LANES=4 UVOL_GEO_CHAIN=0 timeout 300 $B > $O/geo_lanes4_nochain.json 2>> $O/sweep.err
UVOL_GEO_LANES=8 UVOL_GEO_CHAIN=0 timeout 300 $B > $O/geo_lanes8_nochain.json 2>> $O/sweep.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
UVOL_GEO_LANES=1 timeout 600 python bench.py --no-cpu-baseline --no-variants --blocking-calls > $O/bench_lanes1_blocking.json 2>> $O/bench.err
UVOL_GEO_LANES=8 timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_lanes8.json 2>> $O/bench.err
tail -3 $O/pytest.log
