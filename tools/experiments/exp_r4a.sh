# round 4, first run: GPU suite, the default line on distinct-connectivity frames, lanes-per-wave sweeps of the two lane-per-walker kernels
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
for W in 1 2 4 8 16; do
  UVOL_SIMT_W_TRAV=$W timeout 300 python bench.py --only geo --no-variants --no-cpu-baseline --parity-frames 0 --steps 3 > $O/geo_trav_w$W.json 2>> $O/sweep.err
done
for W in 1 2 4 8 32; do
  UVOL_SIMT_W_WALK=$W timeout 300 python bench.py --only geo --no-variants --no-cpu-baseline --parity-frames 0 --steps 3 > $O/geo_walk_w$W.json 2>> $O/sweep.err
done
tail -3 $O/pytest.log
