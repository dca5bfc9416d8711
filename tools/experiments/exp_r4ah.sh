# round 4: host threads per staged upload (UVOL_UP_THREADS 8 = default / 12 / 16 / 4) at the SURVEY 8(d) boundary: texture half alone and the pair
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4ah; mkdir -p $O
for T in 12 16 8 4; do
  UVOL_UP_THREADS=$T timeout 600 python bench.py --host-inputs --no-variants --no-cpu-baseline --steps 3 --warmup 1 --only tex > $O/tex_t$T.json 2> $O/err_t$T.log
  UVOL_UP_THREADS=$T timeout 600 python bench.py --host-inputs --no-variants --no-cpu-baseline --steps 3 --warmup 1 > $O/pair_t$T.json 2>> $O/err_t$T.log
done
nproc > $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>/dev/null
