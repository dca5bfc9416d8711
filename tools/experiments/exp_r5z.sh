# ring shapes, second sweep: more, smaller groups (lanes x groups-per-call; 2560 frames per call)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_z; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "hbm GB", d["config"].get("hbm_in_use_gb_after_timed_steps"), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "seams", round(g.get("geo.k4b_seams",0)), "ent", round(g.get("geo.k7_entropy_encode",0)))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
L() { run l$1_g$2$3 "$4" UVOL_GEO_LANES=$1 UVOL_GEO_GROUPS=$2; }
L 6 4 _a ""
L 8 5 "" ""
L 7 4 "" ""
L 9 6 "" ""
L 12 8 "" ""
L 6 4 _b ""
L 8 6 "" ""
L 3 2 "" ""
L 6 4 _geo "--only geo"
L 3 2 _geo "--only geo"
L 8 5 _geo "--only geo"
