# throughput against frames in flight (compact layout: 31 MB of workspace per frame), geometry alone and the full path
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_d; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --steps 2 --warmup 1 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "ws", d["config"].get("geometry_workspace_bytes_per_frame"), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)), "mism", d.get("parity",{}).get("mismatches"))
except Exception as e: print("$tag FAILED", e)
PY
}
run geo2560 "--only geo" UVOL_TRAV_W=4
run geo3200 "--only geo --frames-per-step 3200" UVOL_TRAV_W=4
run geo3840 "--only geo --frames-per-step 3840" UVOL_TRAV_W=4
run geo4480 "--only geo --frames-per-step 4480" UVOL_TRAV_W=4
run geo3840_3l "--only geo --frames-per-step 3840" UVOL_TRAV_W=4 UVOL_GEO_LANES=3
run full2560 "" UVOL_TRAV_W=4
run full3200 "--frames-per-step 3200" UVOL_TRAV_W=4
