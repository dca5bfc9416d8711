# round 4: what a step of the lane-per-walker kernels waits for: L2 request latency seen by the TCP, address-translation misses,
# L2 hit rate, instruction mix (one group of 1280 frames, one lane, geometry only; one counter set per process)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4s; mkdir -p $O
timeout 120 rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|SQ_[A-Z0-9_]*\|UTCL[A-Z0-9_]*\|TA_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" $O/avail.txt | sort -u > $O/avail_names.txt
i=0
for SET in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCP_UTCL1_PERMISSION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_LRU_INFLIGHT_sum TCP_UTCL1_STALL_MULTI_MISS_sum"; do
  i=$((i+1))
  UVOL_GEO_LANES=1 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o b -- python bench.py --only geo --blocking-calls --no-variants --no-cpu-baseline --parity-frames 0 --steps 1 --warmup 0 --frames-per-step 1280 > /dev/null 2> $O/err_$i.log
  python tools/pmc_any.py $O/p$i $O/set_$i.json simt_f16 entropy_simt k_eb_valence k_renumber > $O/set_$i.txt 2>&1
  rm -rf $O/p$i; tail -c 2000 $O/err_$i.log > $O/e; mv $O/e $O/err_$i.log
done
