# tools/pmc_pack.sh <tag>: HBM traffic of EVERY kernel of one default step, FETCH_SIZE and WRITE_SIZE in their own passes (they do
# not share the TCC slots), the geometry and the texture half in their own processes: with counters armed the runtime serialises
# kernels, and the combined two-stream step once hung in its FETCH_SIZE pass (r02_m, killed by its timeout).
TAG=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for H in geo tex; do for C in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p_${H}_$C -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --only $H > /dev/null 2>> $O/pmc.err
done; done
d() { dirname $(find $O/p_$1 -name bench_counter_collection.csv | head -1); }
python tools/pmc_all.py $(d geo_FETCH_SIZE) $(d geo_WRITE_SIZE) 2160 $O/pmc_all_kernels_geo.json > $O/pmc.log 2>&1
python tools/pmc_all.py $(d tex_FETCH_SIZE) $(d tex_WRITE_SIZE) 2160 $O/pmc_all_kernels_tex.json >> $O/pmc.log 2>&1
python tools/pmc_summary.py $(d geo_FETCH_SIZE) $(d geo_WRITE_SIZE) 2160 $O/pmc_traffic_geo.json >> $O/pmc.log 2>&1
python tools/pmc_summary.py $(d tex_FETCH_SIZE) $(d tex_WRITE_SIZE) 2160 $O/pmc_traffic_tex.json >> $O/pmc.log 2>&1
python - $O <<'P'
import json, sys
o = sys.argv[1]
g, t = json.load(open(o + "/pmc_all_kernels_geo.json")), json.load(open(o + "/pmc_all_kernels_tex.json"))
for r in g["kernels"]: r["half"] = "geometry"
for r in t["kernels"]: r["half"] = "texture"
rows = sorted(g["kernels"] + t["kernels"], key=lambda r: -r["mb_per_frame"])
json.dump({"frames": 2160, "geometry_mb_per_frame": g["total_mb_per_frame"], "texture_mb_per_frame": t["total_mb_per_frame"],
           "total_mb_per_frame": g["total_mb_per_frame"] + t["total_mb_per_frame"], "kernels": rows}, open(o + "/pmc_all_kernels.json", "w"), indent=1)
a, b = json.load(open(o + "/pmc_traffic_geo.json")), json.load(open(o + "/pmc_traffic_tex.json"))
a["kernels"].update(b["kernels"]); json.dump(a, open(o + "/pmc_traffic.json", "w"), indent=1)
print("total %.1f MB/frame (geometry %.1f, texture %.1f)" % (g["total_mb_per_frame"] + t["total_mb_per_frame"], g["total_mb_per_frame"], t["total_mb_per_frame"]))
P
rm -rf $O/p_*
