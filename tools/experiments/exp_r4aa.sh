# round 4: SURVEY 8(d) boundary (host buffers in, bytes in host memory out): blocking passes against enqueued passes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4aa; mkdir -p $O
timeout 600 python bench.py --host-inputs --no-variants --no-cpu-baseline --steps 3 --warmup 1 > $O/host_blocking.json 2> $O/host_blocking.err
timeout 600 python bench.py --host-inputs --host-enqueued --no-variants --no-cpu-baseline --steps 3 --warmup 1 > $O/host_enqueued.json 2> $O/host_enqueued.err
timeout 600 python bench.py --host-inputs --host-enqueued --no-variants --no-cpu-baseline --steps 3 --warmup 1 --only geo > $O/host_enqueued_geo.json 2>> $O/host_enqueued.err
timeout 600 python bench.py --host-inputs --host-enqueued --no-variants --no-cpu-baseline --steps 3 --warmup 1 --only tex > $O/host_enqueued_tex.json 2>> $O/host_enqueued.err
