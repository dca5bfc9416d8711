# three geometry lanes (groups of 1280) beside the texture context: texture call whole / in parts (bounded allocation slack in this build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_o; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "hbm GB", d["config"].get("hbm_in_use_gb_after_timed_steps"), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
run l3_noparts "--steps 4" UVOL_GEO_LANES=3 UVOL_TEX_PART_DEV=0
run l3_parts128 "--steps 4" UVOL_GEO_LANES=3
run l3_parts256 "--steps 4" UVOL_GEO_LANES=3 UVOL_TEX_PART_DEV=256
run l3_noparts_6steps "--steps 6" UVOL_GEO_LANES=3 UVOL_TEX_PART_DEV=0
run l2_noparts "--steps 4" UVOL_TEX_PART_DEV=0
run l3_noparts_w8 "--steps 4" UVOL_GEO_LANES=3 UVOL_TEX_PART_DEV=0 UVOL_TRAV_W=8
run l3_g3 "--steps 4" UVOL_GEO_LANES=3 UVOL_GEO_GROUPS=3 UVOL_TEX_PART_DEV=0
run l3_noparts_3200 "--steps 4 --frames-per-step 3200" UVOL_GEO_LANES=3 UVOL_TEX_PART_DEV=0
