# round 4: one-walker-per-wave kernels on the scalar unit (UVOL_WALK_UNI, default on) against the lane form: kernel durations for one
# group of 1280 / 2560 frames on one lane, then the default line (parity check included) both ways
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4t; mkdir -p $O
for U in 1 0; do for N in 1280 2560; do
  UVOL_WALK_UNI=$U UVOL_GEO_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o b -- python bench.py --only geo --blocking-calls --no-variants --no-cpu-baseline --parity-frames 4 --steps 3 --warmup 1 --frames-per-step $N > $O/line_u${U}_$N.json 2> $O/err_u${U}_$N.log
  cp $(find $O/kt -name b_kernel_stats.csv | head -1) $O/stats_u${U}_$N.csv; rm -rf $O/kt
done; done
for U in 1 0; do
  UVOL_WALK_UNI=$U timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_u$U.json 2> $O/bench_u$U.err
  UVOL_WALK_UNI=$U timeout 600 python bench.py --no-cpu-baseline --no-variants --only geo > $O/bench_geo_u$U.json 2>> $O/bench_u$U.err
done
