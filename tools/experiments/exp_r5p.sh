# small jobs (BASELINE configs[2] = one 300-frame job; the 150-frame share of configs[3]): LDS traversers with the vertex bitmap in L2 when 3 N walkers do not fit otherwise
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_p; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --steps 3 --blocking-calls --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"],1), "trav", round(g.get("geo.k5_traverse",0),1), "walk", round(g.get("geo.k4_eb_walk",0),1), "ent", round(g.get("geo.k7_entropy_encode",0),1), "val", round(g.get("geo.k4_eb_valence",0),1), "seams", round(g.get("geo.k4b_seams",0),1), "corner", round(g.get("geo.k3_corner_table",0),1), "dedup", round(g.get("geo.k2_dedup",0),1))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
for F in 150 300 450 600; do
run j${F}_auto "--frames-per-step $F"
run j${F}_old "--frames-per-step $F" UVOL_TRAV_AUTO_VGLOBAL=0
done
run j300_geo_auto "--frames-per-step 300 --only geo"
run j300_geo_old "--frames-per-step 300 --only geo" UVOL_TRAV_AUTO_VGLOBAL=0
run j300_geo_walkglobal "--frames-per-step 300 --only geo" UVOL_WALK_FORCE=vglobal
