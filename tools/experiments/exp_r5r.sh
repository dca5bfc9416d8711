# sliced front end, confirmation (order mixed: the first run of r05_q was the whole-group form)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_r; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "dedup", round(g.get("geo.k2_dedup",0)), "faces", round(g.get("geo.k2b_faces",0)), "corner", round(g.get("geo.k3_corner_table",0)), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "seams", round(g.get("geo.k4b_seams",0)), "ent", round(g.get("geo.k7_entropy_encode",0)))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
run geo_s128_a "--only geo" UVOL_FE_SLICE=128
run geo_whole_a "--only geo"
run geo_s64 "--only geo" UVOL_FE_SLICE=64
run geo_s256 "--only geo" UVOL_FE_SLICE=256
run geo_whole_b "--only geo"
run geo_s128_b "--only geo" UVOL_FE_SLICE=128
run geo_s128_l2 "--only geo" UVOL_FE_SLICE=128 UVOL_GEO_LANES=2
run geo_whole_l2 "--only geo" UVOL_GEO_LANES=2
run full_s128 "" UVOL_FE_SLICE=128
run full_whole ""
run full_s256 "" UVOL_FE_SLICE=256
