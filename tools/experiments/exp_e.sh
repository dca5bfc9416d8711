cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_e; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python tools/uastc_timing.py 24 > $O/uastc_timing.json 2> $O/err.log
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- $B --only geo --geo-streams 1 --frames-per-step 2160 > $O/geo2160_gs1.json 2>> $O/err.log
cp $(find $O/kt -name bench_kernel_stats.csv | head -1) $O/geo2160_kernel_stats.csv; rm -rf $O/kt
timeout 900 $B --only geo --geo-streams 2 --frames-per-step 2160 > $O/geo2160_gs2.json 2>> $O/err.log
timeout 900 $B --only geo --geo-streams 1 --frames-per-step 240 > $O/geo240.json 2>> $O/err.log
timeout 900 $B --geo-streams 2 --frames-per-step 2160 > $O/full2160_gs2.json 2>> $O/err.log
timeout 900 $B --geo-streams 1 --frames-per-step 2160 > $O/full2160_gs1.json 2>> $O/err.log
timeout 900 $B --geo-streams 2 --frames-per-step 1440 > $O/full1440_gs2.json 2>> $O/err.log
timeout 900 $B > $O/full_default.json 2>> $O/err.log
