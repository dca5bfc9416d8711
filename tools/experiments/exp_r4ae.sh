# round 4: traversers per wave 2 / 3 in the full line (fewer waves: less issue contention between the lanes' launches)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4ae; mkdir -p $O
for W in 2 3 1; do
UVOL_SIMT_W_TRAV=$W timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 > $O/bench_w$W.json 2> $O/bench_w$W.err
done
UVOL_SIMT_W_WALK=1 timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 > $O/bench_walk1.json 2> $O/bench_walk1.err
UVOL_SIMT_W_WALK=4 timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 > $O/bench_walk4.json 2> $O/bench_walk4.err
