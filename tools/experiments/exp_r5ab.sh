# entropy stage under the six-lane ring (640-frame groups): wave-per-stream coder forced, lanes per wave of the lane-per-stream coder
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_ab; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 8 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "ent", round(g.get("geo.k7_entropy_encode",0)), "hist", round(g.get("geo.k7_hist_tables",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "trav", round(g.get("geo.k5_traverse",0)), "parity", d.get("parity",{}).get("mismatches"))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-500:])
PY
}
for i in a b; do
run default_$i ""
run wave_$i "" UVOL_ENTROPY_WAVE=1
run w8_$i "" UVOL_ENTROPY_W=8
run w16_$i "" UVOL_ENTROPY_W=16
done
run w4 "" UVOL_ENTROPY_W=4
run w64 "" UVOL_ENTROPY_W=64
# walkers per wave once more, in the six-lane mix
run walk_w4 "" UVOL_SIMT_W_WALK=4
run walk_w64 "" UVOL_SIMT_W_WALK=64
run trav_w2 "" UVOL_TRAV_W=2
run trav_w8 "" UVOL_TRAV_W=8
run default_c ""
