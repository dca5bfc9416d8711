# round 4: lanes per context after the traversal fix (2 / 3 / 4), geometry alone and the full line; texture alone
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4x; mkdir -p $O
for L in 2 3 4; do
  UVOL_GEO_LANES=$L timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 > $O/bench_l$L.json 2> $O/bench_l$L.err
  UVOL_GEO_LANES=$L timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 --only geo > $O/geo_l$L.json 2>> $O/bench_l$L.err
done
timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 --only tex > $O/tex.json 2> $O/tex.err
UVOL_GEO_LANES=4 UVOL_GEO_MIN_GROUP=320 timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 --frames-per-step 1280 > $O/bench_l4_1280.json 2> $O/bench_l4_1280.err
