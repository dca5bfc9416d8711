# round 4: the walkers' step as ONE loop per run of steps (one exit, the rare path outside): kernel durations for one group of 1280 / 2560
# frames on one lane, the default line (parity check included), geometry alone; GPU geometry tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4ai; mkdir -p $O
for N in 1280 2560; do
  UVOL_GEO_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o b -- python bench.py --only geo --blocking-calls --no-variants --no-cpu-baseline --parity-frames 4 --steps 3 --warmup 1 --frames-per-step $N > $O/line_$N.json 2> $O/err_$N.log
  cp $(find $O/kt -name b_kernel_stats.csv | head -1) $O/stats_$N.csv; rm -rf $O/kt
done
timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench2.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-variants --only geo > $O/bench_geo.json 2>> $O/bench.err
timeout 1200 python -m pytest tests/test_gpu_geom.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
