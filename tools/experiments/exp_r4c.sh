# round 4: do the lanes' streams share hardware queues?  The same sweep with more HW queues than streams.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
B="python bench.py --only geo --no-variants --no-cpu-baseline --parity-frames 0 --steps 4"
for Q in 8 24; do for L in 1 2 4 8; do
  GPU_MAX_HW_QUEUES=$Q UVOL_GEO_LANES=$L timeout 300 $B > $O/geo_q${Q}_lanes${L}.json 2>> $O/sweep.err
done; done
GPU_MAX_HW_QUEUES=24 UVOL_GEO_LANES=4 UVOL_GEO_CHAIN=0 timeout 300 $B > $O/geo_q24_lanes4_nochain.json 2>> $O/sweep.err
GPU_MAX_HW_QUEUES=24 UVOL_GEO_LANES=4 timeout 300 $B --blocking-calls > $O/geo_q24_lanes4_blocking.json 2>> $O/sweep.err
GPU_MAX_HW_QUEUES=24 UVOL_GEO_LANES=4 timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_q24_lanes4.json 2>> $O/bench.err
GPU_MAX_HW_QUEUES=24 UVOL_GEO_LANES=1 timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_q24_lanes1.json 2>> $O/bench.err
# one record per face (format 2) against the 8-byte corner records in the lane-per-walker kernels, one lane
GPU_MAX_HW_QUEUES=24 UVOL_GEO_LANES=1 UVOL_REC_FACE=0 timeout 300 $B > $O/geo_q24_lanes1_cornerrec.json 2>> $O/sweep.err
timeout 900 python -m pytest tests/test_gpu_geom.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
