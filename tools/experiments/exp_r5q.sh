# the streaming front end launched slice by slice (intermediates of a slice stay in the Infinity Cache?) - judged by time; new defaults (3 lanes, texture parts, 6 steps)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_q; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "hbm GB", d["config"].get("hbm_in_use_gb_after_timed_steps"), "dedup", round(g.get("geo.k2_dedup",0)), "faces", round(g.get("geo.k2b_faces",0)), "corner", round(g.get("geo.k3_corner_table",0)), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
run geo_whole "--only geo"
run geo_s8 "--only geo" UVOL_FE_SLICE=8
run geo_s32 "--only geo" UVOL_FE_SLICE=32
run geo_s128 "--only geo" UVOL_FE_SLICE=128
run full_whole ""
run full_s8 "" UVOL_FE_SLICE=8
run full_s32 "" UVOL_FE_SLICE=32
run full_s128 "" UVOL_FE_SLICE=128
run full_whole_again ""
