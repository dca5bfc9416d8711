# front-end chain placed after the mid-batch read-back (UVOL_GEO_CHAIN=3): the host waits for this group's dedup only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_aa; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; args=$1; shift; env "$@" UVOL_TIMING=1 timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json, re
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    rows=[list(map(float,re.findall(r"host prepared ([\d.]+) ms, enqueued ([\d.]+), gpu done ([\d.]+), packed d2h ([\d.]+), copied out ([\d.]+)",l)[0])) for l in open("$O/$tag.err") if "geo group n=" in l][4:]
    m=[sum(r[k] for r in rows)/max(1,len(rows)) for k in range(5)]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "| per group: enqueued %.0f gpu done %.0f copied %.0f ms" % (m[1], m[2], m[4]), "dedup", round(g.get("geo.k2_dedup",0)), "corner", round(g.get("geo.k3_corner_table",0)), "walk", round(g.get("geo.k4_eb_walk",0)), "trav", round(g.get("geo.k5_traverse",0)))
except Exception as e: print("$tag FAILED", e)
PY
rm -f $O/$tag.err
}
run chain1_a ""
run chain3_a "" UVOL_GEO_CHAIN=3
run chain1_b ""
run chain3_b "" UVOL_GEO_CHAIN=3
run chain0 "" UVOL_GEO_CHAIN=0
run chain3_geo "--only geo" UVOL_GEO_CHAIN=3
run chain1_geo "--only geo"
