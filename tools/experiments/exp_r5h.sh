# cooperative per-face-record walkers at mid sizes (their per-step time depends on the waves resident per CU)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_h; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --steps 3 --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)))
except Exception as e: print("$tag FAILED", e)
PY
}
for F in 320 640 1280; do
run f${F}_default "--only geo --frames-per-step $F --blocking-calls"
run f${F}_wave4 "--only geo --frames-per-step $F --blocking-calls" UVOL_SIMT_W=1 UVOL_TRAV_W=4 UVOL_SIMT_W_WALK=16
run f${F}_coop "--only geo --frames-per-step $F --blocking-calls" UVOL_SIMT_W=1 UVOL_TRAV_FORM=coop UVOL_WALK_FORM=coop
done
