# round 4: issue priority of the walkers' waves (s_setprio 0 / 1 / 3) in the full line and geometry alone
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4an; mkdir -p $O
for P in 3 0 1; do
  UVOL_WALK_PRIO=$P timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 > $O/bench_p$P.json 2> $O/bench_p$P.err
  UVOL_WALK_PRIO=$P timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 --only geo > $O/geo_p$P.json 2>> $O/bench_p$P.err
done
