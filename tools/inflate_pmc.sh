# tools/inflate_pmc.sh <tag>: instruction and wait counters of k_inflate (one launch of 240 streams), one rocprofv3 --pmc pass per group
TAG=$1; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
i=0
for C in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
  i=$((i+1)); rm -rf $O/p$i
  timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o inf -- python tools/inflate_timing.py 240 > /dev/null 2> $O/p$i.err
  python tools/pmc_any.py $O/p$i $O/pmc_$i.json k_inflate >> $O/pmc.log 2>&1
  rm -rf $O/p$i
done
cat $O/pmc.log
