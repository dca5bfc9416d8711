# round 4: (1) default line with the variants (after the context renewal); (2) how each geometry kernel's duration grows with the frames
# of ONE launch (one lane, one blocking call per pass): 160 ... 2560 frames
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4r; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
for N in 160 320 640 1280 2560; do
  UVOL_GEO_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$N -o b -- python bench.py --only geo --blocking-calls --no-variants --no-cpu-baseline --parity-frames 0 --steps 3 --warmup 1 --frames-per-step $N > $O/line_$N.json 2> $O/err_$N.log
  cp $(find $O/kt$N -name b_kernel_stats.csv | head -1) $O/stats_$N.csv; rm -rf $O/kt$N
done
