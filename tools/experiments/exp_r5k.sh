# frames in flight: the walkers' chain is flat in the frame count, so throughput should follow the frames a pass holds until the issue slots fill (workspace 31 MB per frame)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_k; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --steps 2 --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "ent", round(g.get("geo.k7_entropy_encode",0)))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
run geo2560_2l "--only geo"
run geo3840_2l "--only geo --frames-per-step 3840"
run geo3840_3l "--only geo --frames-per-step 3840" UVOL_GEO_LANES=3
run geo5120_4l "--only geo --frames-per-step 5120" UVOL_GEO_LANES=4
run geo5120_2l "--only geo --frames-per-step 5120"
run geo3840_3l_w8 "--only geo --frames-per-step 3840" UVOL_GEO_LANES=3 UVOL_TRAV_W=8
run full3840_3l "--frames-per-step 3840" UVOL_GEO_LANES=3
run full3200_2l "--frames-per-step 3200"
