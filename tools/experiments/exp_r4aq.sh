# round 4: how the groups of a pass overlap on the lanes (kernel trace of geometry alone, steady-state passes) -> tools/lane_overlap.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4aq; mkdir -p $O
for L in 2 1; do
  UVOL_GEO_LANES=$L timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt$L -o b -- python bench.py --only geo --no-variants --no-cpu-baseline --parity-frames 0 --steps 3 --warmup 1 > $O/line_l$L.json 2> $O/err_l$L.log
  python tools/lane_overlap.py $(find $O/kt$L -name b_kernel_trace.csv | head -1) > $O/overlap_l$L.json 2>> $O/err_l$L.log
  rm -rf $O/kt$L
done
