# round 4: walkers' loads as agent-scope atomics (UVOL_WALK_LD=1: no L1 look-up) against plain loads; one group of 1280 / 2560 frames on one lane
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4u; mkdir -p $O
for U in 1 0; do for N in 1280 2560; do
  UVOL_WALK_LD=$U UVOL_GEO_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o b -- python bench.py --only geo --blocking-calls --no-variants --no-cpu-baseline --parity-frames 4 --steps 3 --warmup 1 --frames-per-step $N > $O/line_u${U}_$N.json 2> $O/err_u${U}_$N.log
  cp $(find $O/kt -name b_kernel_stats.csv | head -1) $O/stats_u${U}_$N.csv; rm -rf $O/kt
done; done
