# round 4: PC sampling of the attribute traversal (where a walker's wave sits): one group of 640 frames, geometry only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4v; mkdir -p $O
for M in "stochastic cycles 65536" "host_trap time 20"; do
  set -- $M
  UVOL_GEO_LANES=1 timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --kernel-trace --output-format csv -d $O/p_$1 -o b -- python bench.py --only geo --blocking-calls --no-variants --no-cpu-baseline --parity-frames 0 --steps 1 --warmup 0 --frames-per-step 640 > $O/line_$1.json 2> $O/err_$1.log
  echo "rc $?" >> $O/err_$1.log
  python tools/pcs_reduce.py $O/p_$1 $O/trav_$1.txt k_traverse_simt_f16 > $O/red_$1.log 2>&1
  python tools/pcs_reduce.py $O/p_$1 $O/walk_$1.txt k_eb_walk_simt_f16 >> $O/red_$1.log 2>&1
  ls -la $O/p_$1/* >> $O/red_$1.log 2>&1; find $O/p_$1 -name "*.csv" | xargs ls -la >> $O/red_$1.log 2>&1
  rm -rf $O/p_$1; tail -c 3000 $O/err_$1.log > $O/e; mv $O/e $O/err_$1.log
done
