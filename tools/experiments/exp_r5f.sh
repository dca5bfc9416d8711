# edgebreaker walk: lane form (16 per wave, idle lanes leave) against the wave form (pops as steps, outputs through LDS rings)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_f; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --steps 2 --warmup 1 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)), "mism", d.get("parity",{}).get("mismatches"))
except Exception as e: print("$tag FAILED", e)
PY
}
run walk_lane "--only geo" UVOL_TRAV_W=4 UVOL_WALK_FORM=lane
run walk_wave2 "--only geo" UVOL_TRAV_W=4 UVOL_WALK_W=2
run walk_wave4 "--only geo" UVOL_TRAV_W=4 UVOL_WALK_W=4
run walk_wave8 "--only geo" UVOL_TRAV_W=4 UVOL_WALK_W=8
run walk_wave16 "--only geo" UVOL_TRAV_W=4 UVOL_WALK_W=16
run walk_lane1 "--only geo" UVOL_TRAV_W=4 UVOL_WALK_FORM=lane UVOL_SIMT_W_WALK=1
