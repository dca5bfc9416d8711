# stagger of the groups on the ring: front end behind the previous group's front end (1, default) or behind its WALK (2); geometry alone is bimodal with 3 lanes (3500 / 4800)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "dedup", round(g.get("geo.k2_dedup",0)), "corner", round(g.get("geo.k3_corner_table",0)), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "seams", round(g.get("geo.k4b_seams",0)), "ent", round(g.get("geo.k7_entropy_encode",0)))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
for i in 1 2 3; do
run geo_chain1_$i "--only geo"
run geo_chain2_$i "--only geo" UVOL_GEO_CHAIN=2
done
run geo_chain0 "--only geo" UVOL_GEO_CHAIN=0
run geo_chain2_l4 "--only geo" UVOL_GEO_CHAIN=2 UVOL_GEO_LANES=4
for i in 1 2; do
run full_chain1_$i ""
run full_chain2_$i "" UVOL_GEO_CHAIN=2
done
run full_chain0 "" UVOL_GEO_CHAIN=0
