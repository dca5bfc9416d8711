cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_k; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout 900 $B > $O/full_default.json 2> $O/err.log
timeout 900 $B --tex-priority 0 > $O/full_tp0.json 2>> $O/err.log
timeout 900 $B --tex-priority 0 --geo-priority 1 > $O/full_tp0_gp1.json 2>> $O/err.log
timeout 900 $B --geo-streams 2 --tex-priority 0 --geo-priority 1 > $O/full_gs2_tp0_gp1.json 2>> $O/err.log
timeout 900 $B --only geo > $O/geo.json 2>> $O/err.log
