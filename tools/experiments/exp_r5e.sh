# wave traverser with its stores off the critical path (order[] through an LDS ring, predicated pending / vertex-word stores)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_e; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --steps 2 --warmup 1 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)), "mism", d.get("parity",{}).get("mismatches"))
except Exception as e: print("$tag FAILED", e)
PY
}
run geo_w2 "--only geo" UVOL_TRAV_W=2
run geo_w4 "--only geo" UVOL_TRAV_W=4
run geo_w8 "--only geo" UVOL_TRAV_W=8
run geo_w16 "--only geo" UVOL_TRAV_W=16
run geo3200_w4 "--only geo --frames-per-step 3200" UVOL_TRAV_W=4
run tex "--only tex"
run full_w4 "" UVOL_TRAV_W=4
