# instruction mix and L2 behaviour of the serial kernels on the round's final build (one group of 1280 distinct frames, one lane, geometry only; one counter set per process)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_v; rm -rf $O; mkdir -p $O
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  for TRY in 1 2; do
    rm -rf $O/p$i
    UVOL_GEO_LANES=1 timeout 240 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -o b -- python bench.py --only geo --blocking-calls --no-variants --no-cpu-baseline --parity-frames 0 --steps 1 --warmup 0 --frames-per-step 1280 > /dev/null 2> $O/err_$i.log
    if python tools/pmc_any.py $O/p$i $O/set_$i.json traverse_wave walk_simt_f16 entropy_simt k_eb_valence > $O/set_$i.txt 2>&1; then break; fi
  done
  rm -rf $O/p$i; tail -c 1500 $O/err_$i.log > $O/e; mv $O/e $O/err_$i.log
done
python - <<PY
import json, glob
m = {}
for f in sorted(glob.glob("$O/set_*.json")):
    for k, v in json.load(open(f)).items(): m.setdefault(k, {}).update(v)
steps = {"k_traverse_wave_f16": 3 * 1280 * 200100.0, "k_eb_walk_simt_f16": 1280 * 200100.0}
out = {"what": "rocprofv3 --pmc passes of one geometry group (1280 distinct frames, one lane; tools/experiments/exp_r5v.sh); raw counter sums per launch and, for the walkers, per walker step (faces x tables x frames; the wave form's instructions are shared by the 4 walkers of a wave, the walk's by 16)", "kernels": m, "per_walker_step": {}}
for k, n in steps.items():
    for kk, v in m.items():
        if kk.startswith(k):
            out["per_walker_step"][kk] = {c: round(x / n, 3) for c, x in v.items() if c != "launches"}
json.dump(out, open("$O/walker_counters.json", "w"), indent=1)
print(json.dumps(out["per_walker_step"], indent=1))
PY
