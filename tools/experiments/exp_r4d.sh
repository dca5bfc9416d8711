# round 4: lanes again, after the job-record read-back left geo_submit (it blocked the host until the group was done)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
B="python bench.py --only geo --no-variants --no-cpu-baseline --parity-frames 0 --steps 4"
for L in 1 2 4 8; do
  UVOL_GEO_LANES=$L timeout 300 $B > $O/geo_lanes${L}.json 2>> $O/sweep.err
done
UVOL_GEO_LANES=4 timeout 300 $B --blocking-calls > $O/geo_lanes4_blocking.json 2>> $O/sweep.err
UVOL_GEO_LANES=4 UVOL_GEO_CHAIN=0 timeout 300 $B > $O/geo_lanes4_nochain.json 2>> $O/sweep.err
GPU_MAX_HW_QUEUES=24 UVOL_GEO_LANES=4 timeout 300 $B > $O/geo_q24_lanes4.json 2>> $O/sweep.err
GPU_MAX_HW_QUEUES=24 UVOL_GEO_LANES=8 timeout 300 $B > $O/geo_q24_lanes8.json 2>> $O/sweep.err
UVOL_GEO_LANES=4 timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_lanes4.json 2>> $O/bench.err
GPU_MAX_HW_QUEUES=24 UVOL_GEO_LANES=4 timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_q24_lanes4.json 2>> $O/bench.err
UVOL_GEO_LANES=8 timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_lanes8.json 2>> $O/bench.err
