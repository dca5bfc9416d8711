# tools/pmc_pack.sh <tag>: HBM traffic of EVERY kernel of one default step.  FETCH_SIZE and WRITE_SIZE in their own passes (they do
# not share the TCC slots), the geometry and the texture half in their own processes, every launch synchronised (UVOL_DEBUG=1):
# with counters armed the runtime serialises kernels and the two-stream step deadlocks on its cross-stream event waits.  On this
# pool a pass sometimes hangs right after tool initialisation; each pass is therefore tried up to three times under a short timeout
# and reduced to a per-kernel JSON at once (tools/pmc_one.py); tools/pmc_merge.py builds the tables from whatever passes exist.
TAG=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
: > $O/pmc.log
for H in ${PMC_HALVES:-geo tex}; do for C in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  for TRY in 1 2 3; do
    rm -rf $O/p_${H}_$C
    UVOL_GEO_LANES=1 UVOL_DEBUG=1 timeout ${PMC_TIMEOUT:-100} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p_${H}_$C -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames-per-step 2160 --only $H > /dev/null 2> $O/pmc_${H}_$C.err
    if python tools/pmc_one.py $O/p_${H}_$C $C $O/pmc_${H}_$C.json >> $O/pmc.log 2>&1; then echo "$H $C ok (try $TRY)" >> $O/pmc.log; break; else echo "$H $C FAILED (try $TRY)" >> $O/pmc.log; fi
  done
  rm -rf $O/p_${H}_$C $O/pmc_${H}_$C.err
done; done
python tools/pmc_merge.py $O >> $O/pmc.log 2>&1
