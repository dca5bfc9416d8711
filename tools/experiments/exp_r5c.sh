# does geometry throughput follow the frames in flight?  (the serial chain's time is flat in the frame count); walkers per wave in the full mix
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 600 python bench.py $args --no-variants --no-cpu-baseline --steps 2 --warmup 1 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)), "mism", d.get("parity",{}).get("mismatches"))
except Exception as e: print("$tag FAILED", e)
PY
}
run geo2560 "--only geo" UVOL_TRAV_W=4
run geo3072 "--only geo --frames-per-step 3072" UVOL_TRAV_W=4
run geo3584 "--only geo --frames-per-step 3584" UVOL_TRAV_W=4
run geo3584_3l "--only geo --frames-per-step 3584" UVOL_TRAV_W=4 UVOL_GEO_LANES=3
run full_w2 "" UVOL_TRAV_W=2
run full_w4 "" UVOL_TRAV_W=4
run full_w8 "" UVOL_TRAV_W=8
