# walkers per wave in the six-lane mix, confirmation
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_ac; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "ent", round(g.get("geo.k7_entropy_encode",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "trav", round(g.get("geo.k5_traverse",0)))
except Exception as e: print("$tag FAILED", e)
PY
}
for i in a b c; do
run trav8_$i "" UVOL_TRAV_W=8
run default_$i ""
run trav8_walk4_$i "" UVOL_TRAV_W=8 UVOL_SIMT_W_WALK=4
done
run trav6 "" UVOL_TRAV_W=6
run trav12 "" UVOL_TRAV_W=12
run trav8_geo "--only geo" UVOL_TRAV_W=8
run default_geo "--only geo"
