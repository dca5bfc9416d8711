# tools/pmc_pack.sh <tag>: HBM traffic of EVERY kernel of one default step, FETCH_SIZE and WRITE_SIZE in their own passes (they do
# not share the TCC slots), the geometry and the texture half in their own processes: with counters armed the runtime serialises
# kernels, and the combined two-stream step once hung in its FETCH_SIZE pass (r02_m, killed by its timeout).
TAG=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
# PMC_HALVES="geo" / "tex" restricts the passes; PMC_DEBUG=1 runs them with UVOL_DEBUG=1 (every launch named on stderr and
# synchronised), which shows the kernel a hanging pass stopped in
for H in ${PMC_HALVES:-geo tex}; do for C in FETCH_SIZE WRITE_SIZE; do
  UVOL_DEBUG=${PMC_DEBUG:-0} timeout ${PMC_TIMEOUT:-420} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p_${H}_$C -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --only $H > /dev/null 2> $O/pmc_${H}_$C.err
  grep '^\[uvol\]' $O/pmc_${H}_$C.err | tail -3 > $O/pmc_${H}_$C.last; grep -v '^\[uvol\]\|amdgpu.ids' $O/pmc_${H}_$C.err >> $O/pmc.err; mv $O/pmc_${H}_$C.err $O/pmc_${H}_$C.errfull; tail -c 3000 $O/pmc_${H}_$C.errfull > $O/pmc_${H}_$C.tail; rm -f $O/pmc_${H}_$C.errfull
done; done
d() { dirname $(find $O/p_$1 -name bench_counter_collection.csv | head -1); }
: > $O/pmc.log
for H in ${PMC_HALVES:-geo tex}; do
  python tools/pmc_all.py $(d ${H}_FETCH_SIZE) $(d ${H}_WRITE_SIZE) 2160 $O/pmc_all_kernels_$H.json >> $O/pmc.log 2>&1
  python tools/pmc_summary.py $(d ${H}_FETCH_SIZE) $(d ${H}_WRITE_SIZE) 2160 $O/pmc_traffic_$H.json >> $O/pmc.log 2>&1
done
python tools/pmc_merge.py $O >> $O/pmc.log 2>&1
rm -rf $O/p_*
