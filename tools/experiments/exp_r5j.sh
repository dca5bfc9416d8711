# XCD census (which XCD does CU-mask bit i select?) and the co-run with the texture / geometry contexts on disjoint sets of XCDs; pinned host inputs after the upload-gate fix
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_j; rm -rf $O; mkdir -p $O
timeout 120 tools/xcd/xcd_census > $O/xcd_census.json 2> $O/xcd_census.err; head -c 3000 $O/xcd_census.json
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --steps 2 --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)), "fit", round(g.get("tex.k9_endpoint_fit",0)), "selcb", round(g.get("tex.k10_selector_codebook",0)))
except Exception as e: print("$tag FAILED", e)
PY
}
run base ""
run t2_g6 "--tex-cus 8:03 --geo-cus 8:fc"
run t3_g5 "--tex-cus 8:07 --geo-cus 8:f8"
run t4_g4 "--tex-cus 8:0f --geo-cus 8:f0"
run t2_gall "--tex-cus 8:03"
run t4_gall "--tex-cus 8:0f"
run tall_g6 "--geo-cus 8:fc"
run host_pinned "--host-inputs --host-pinned"
run host "--host-inputs"
