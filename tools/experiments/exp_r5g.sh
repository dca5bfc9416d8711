# cooperative traverser on per-face records (one walker per wave, state in SGPRs) against the wave form (4 walkers per wave)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_g; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --steps 2 --warmup 1 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)), "mism", d.get("parity",{}).get("mismatches"))
except Exception as e: print("$tag FAILED", e)
PY
}
run wave4 "--only geo" UVOL_TRAV_W=4
run coop "--only geo" UVOL_TRAV_FORM=coop
run coop_1lane "--only geo" UVOL_TRAV_FORM=coop UVOL_GEO_LANES=1
run wave4_1lane "--only geo" UVOL_TRAV_W=4 UVOL_GEO_LANES=1
run coop_full "" UVOL_TRAV_FORM=coop
run coop3200 "--only geo --frames-per-step 3200" UVOL_TRAV_FORM=coop
run coopboth "--only geo" UVOL_TRAV_FORM=coop UVOL_WALK_FORM=coop
run coopboth_full "" UVOL_TRAV_FORM=coop UVOL_WALK_FORM=coop
run coopboth3200 "--only geo --frames-per-step 3200" UVOL_TRAV_FORM=coop UVOL_WALK_FORM=coop
