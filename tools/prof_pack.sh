# tools/prof_pack.sh <tag>: the measurement pack of a build (run on the GPU box through gpurun); results under gpurun_out/<tag>/
TAG=$1
ulimit -c 0; export HSA_ENABLE_COREDUMP=0          # a GPU fault must not fill the box's disk with a core dump (it did once, and every later step failed)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench_run2.json 2>> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $O/bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/kt -name bench_kernel_stats.csv | head -1) $O/kernel_stats.csv
# HBM traffic of every kernel: tools/pmc_pack.sh (own processes per half)
rm -rf $O/kt
PMC_TIMEOUT=200 bash tools/pmc_pack.sh $TAG
python tools/uastc_timing.py 24 > $O/uastc_timing.json 2>> $O/bench.err
python tools/dec_timing.py 96 > $O/dec_timing.json 2>> $O/bench.err
python tools/gdec_timing.py 1920 > $O/gdec_timing.json 2>> $O/bench.err
bash tools/variants.sh $TAG >> $O/bench.err 2>&1
