# ring of lanes longer than the groups of a call: consecutive enqueued passes overlap (frames in flight = lanes x group size, inputs of ONE pass resident)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_m; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "hbm GB", d["config"].get("hbm_in_use_gb_after_timed_steps"), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
run geo_l2 "--only geo --steps 4"
run geo_l3 "--only geo --steps 4" UVOL_GEO_LANES=3
run geo_l4 "--only geo --steps 4" UVOL_GEO_LANES=4
run geo_l3_g1 "--only geo --steps 4" UVOL_GEO_LANES=3 UVOL_GEO_GROUPS=1
run full_l2 "--steps 4"
run full_l3 "--steps 4" UVOL_GEO_LANES=3
run full_l4 "--steps 4" UVOL_GEO_LANES=4
run full_l3_w8 "--steps 4" UVOL_GEO_LANES=3 UVOL_TRAV_W=8
run full_l3_3steps "--steps 3" UVOL_GEO_LANES=3
# CU partition INSIDE every XCD (the only partition a CU mask can express: profiles/r05_xcd_census.json): texture on the first k CUs of each XCD, geometry on the rest
run full_t8_g24 "--steps 4 --tex-cus xcd:0-8 --geo-cus xcd:8-32"
run full_t12_g20 "--steps 4 --tex-cus xcd:0-12 --geo-cus xcd:12-32"
run full_t8_gall "--steps 4 --tex-cus xcd:0-8"
run full_t16_gall "--steps 4 --tex-cus xcd:0-16"
