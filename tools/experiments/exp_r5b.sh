# wave-form traverser (k_traverse_wave_f16) against the lane form, walkers per wave swept; geometry alone, 2560 distinct frames per step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_b; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --only geo --no-variants --no-cpu-baseline --steps 2 --warmup 1 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)), "parity", d.get("parity"))
except Exception as e: print("$tag FAILED", e)
PY
}
run lane UVOL_TRAV_FORM=lane
run wave1 UVOL_TRAV_W=1
run wave2 UVOL_TRAV_W=2
run wave4 UVOL_TRAV_W=4
run wave8 UVOL_TRAV_W=8
run wave16 UVOL_TRAV_W=16
run wave32 UVOL_TRAV_W=32
run wave8_3lanes UVOL_TRAV_W=8 UVOL_GEO_LANES=3
run wave8_4lanes UVOL_TRAV_W=8 UVOL_GEO_LANES=4
