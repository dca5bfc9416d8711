# per-kernel HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the geometry path at 2160 frames per launch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_f; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --steps 1 --warmup 0 --only geo --geo-streams 1 --frames-per-step 2160"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o bench -- $B > /dev/null 2> $O/err.log
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o bench -- $B > /dev/null 2>> $O/err.log
PF=$(dirname $(find $O/pf -name bench_counter_collection.csv | head -1)); PW=$(dirname $(find $O/pw -name bench_counter_collection.csv | head -1))
python tools/pmc_all.py $PF $PW 2160 $O/pmc_all_geo2160.json > $O/pmc.log 2>&1
rm -rf $O/pf $O/pw
