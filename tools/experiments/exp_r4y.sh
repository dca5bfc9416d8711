# round 4: stream priorities of the two halves after the lane work (texture high / geometry high / both / neither)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4y; mkdir -p $O
for T in 1 0; do for G in 0 1; do
  timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 --tex-priority $T --geo-priority $G > $O/bench_t${T}_g$G.json 2> $O/bench_t${T}_g$G.err
done; done
timeout 600 python tools/gdec_timing.py 1920 > $O/gdec_timing.json 2> $O/gdec.err
