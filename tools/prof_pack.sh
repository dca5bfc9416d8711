TAG=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench_run2.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --host-inputs > $O/bench_host_inputs.json 2>> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2>> $O/bench.err
cp $O/kt/*/bench_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $O/kt/bench_kernel_stats.csv $O/kernel_stats.csv
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>> $O/bench.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>> $O/bench.err
PF=$(dirname $(find $O/pf -name bench_counter_collection.csv | head -1)); PW=$(dirname $(find $O/pw -name bench_counter_collection.csv | head -1))
python tools/pmc_summary.py $PF $PW 240:720 $O/pmc_traffic.json > $O/pmc.log 2>&1
rm -rf $O/kt $O/pf $O/pw
