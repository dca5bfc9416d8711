# round 4: mid-size jobs (300 / 450 frames, blocking): lane-per-walker traversers against LDS traversers with the vertex bitmap in L2 (six per CU)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4ao; mkdir -p $O
for N in 300 450 600; do for V in 0 1; do
  timeout 600 python bench.py --no-cpu-baseline --no-variants --steps 3 --warmup 1 --frames-per-step $N --blocking-calls --traverse-vbits-l2 $V > $O/job_${N}_v$V.json 2> $O/err_${N}_v$V.log
  timeout 600 python bench.py --no-cpu-baseline --no-variants --steps 3 --warmup 1 --frames-per-step $N --blocking-calls --traverse-vbits-l2 $V --only geo > $O/geo_${N}_v$V.json 2>> $O/err_${N}_v$V.log
done; done
