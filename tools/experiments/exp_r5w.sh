# ring shapes: more, smaller groups (a spare lane hides the host's complete -> submit gap; the walkers' chain is flat in the group size down to ~640 frames)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${WTAG:-r05_w}; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "hbm GB", d["config"].get("hbm_in_use_gb_after_timed_steps"), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "seams", round(g.get("geo.k4b_seams",0)), "ent", round(g.get("geo.k7_entropy_encode",0)))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
run l3_g2 ""
run l4_g3 "" UVOL_GEO_LANES=4 UVOL_GEO_GROUPS=3
run l5_g4 "" UVOL_GEO_LANES=5 UVOL_GEO_GROUPS=4
run l5_g3 "" UVOL_GEO_LANES=5 UVOL_GEO_GROUPS=3
run l6_g4 "" UVOL_GEO_LANES=6 UVOL_GEO_GROUPS=4
run l4_g2_1920 "--frames-per-step 1920" UVOL_GEO_LANES=4
run l3_g2_12steps "--steps 12"
run l4_g3_12steps "--steps 12" UVOL_GEO_LANES=4 UVOL_GEO_GROUPS=3
