# round 4: traversal with the batched component-start scan: lanes per wave 1 / 4 / 16 (UVOL_SIMT_W_TRAV), one group of 1280 / 2560 frames on one lane;
# then the default line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4w; mkdir -p $O
for W in 1 4 16; do for N in 1280 2560; do
  UVOL_SIMT_W_TRAV=$W UVOL_GEO_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o b -- python bench.py --only geo --blocking-calls --no-variants --no-cpu-baseline --parity-frames 4 --steps 3 --warmup 1 --frames-per-step $N > $O/line_w${W}_$N.json 2> $O/err_w${W}_$N.log
  cp $(find $O/kt -name b_kernel_stats.csv | head -1) $O/stats_w${W}_$N.csv; rm -rf $O/kt
done; done
for W in 1 4 16; do
UVOL_SIMT_W_TRAV=$W timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_w$W.json 2> $O/bench_w$W.err
done
