# timeline of the full default step (kernel trace only): where do the geometry lanes wait beside the texture context?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_n; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --parity-frames 0 $BENCH_ARGS > $O/bench.json 2> $O/bench.err
F=$(find $O/kt -name "bench_kernel_trace.csv" | head -1)
TL_THRESH_MS=4 python tools/trace_timeline.py $F > $O/timeline.txt 2>&1
python tools/lane_overlap.py $F > $O/lane_overlap.json 2>&1
rm -rf $O/kt
head -c 400 $O/bench.json; echo; head -150 $O/timeline.txt
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "hbm GB", d["config"].get("hbm_in_use_gb_after_timed_steps"), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
# texture calls on device inputs in parts of 128 segments (40 GB less workspace): room for a third geometry lane?
run full_l2_parts "--steps 4"
run full_l2_noparts "--steps 4" UVOL_TEX_PART_DEV=0
run full_l3_parts "--steps 4" UVOL_GEO_LANES=3
run full_l3_parts64 "--steps 4" UVOL_GEO_LANES=3 UVOL_TEX_PART_DEV=64
run tex_parts "--only tex --steps 4"
run tex_noparts "--only tex --steps 4" UVOL_TEX_PART_DEV=0
