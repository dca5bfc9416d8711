# round 4: which lane count for the default?  Full line with variants (small jobs, host inputs) per lane count
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
for L in 1 4; do
  UVOL_GEO_LANES=$L timeout 900 python bench.py --no-cpu-baseline > $O/bench_lanes$L.json 2>> $O/bench.err
done
UVOL_GEO_LANES=2 timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_lanes2.json 2>> $O/bench.err
UVOL_GEO_LANES=3 timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_lanes3.json 2>> $O/bench.err
UVOL_GEO_LANES=2 timeout 300 python bench.py --only geo --no-variants --no-cpu-baseline --parity-frames 0 --steps 4 > $O/geo_lanes2.json 2>> $O/bench.err
UVOL_GEO_LANES=3 timeout 300 python bench.py --only geo --no-variants --no-cpu-baseline --parity-frames 0 --steps 4 > $O/geo_lanes3.json 2>> $O/bench.err
