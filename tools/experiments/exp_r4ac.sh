# round 4: frames per step (frames in flight): 2560 (default) / 3072 / 3328; a group is half of it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4ac; mkdir -p $O
for F in 3072 3330 2560; do
  timeout 700 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 --frames-per-step $F > $O/bench_f$F.json 2> $O/bench_f$F.err
  timeout 700 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 --frames-per-step $F --only geo > $O/geo_f$F.json 2>> $O/bench_f$F.err
done
