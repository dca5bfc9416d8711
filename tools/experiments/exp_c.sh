cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_c; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --only geo --geo-streams 1 --frames-per-step 2160 > $O/geo2160_gs1.json 2> $O/err.log
cp $(find $O/kt -name bench_kernel_stats.csv | head -1) $O/geo2160_kernel_stats.csv; rm -rf $O/kt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --only tex --frames-per-step 2160 > $O/tex2160.json 2>> $O/err.log
cp $(find $O/kt -name bench_kernel_stats.csv | head -1) $O/tex2160_kernel_stats.csv; rm -rf $O/kt
