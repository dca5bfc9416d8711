cd $GRAFT_REPO_ROOT; O=gpurun_out/r3f; rm -rf $O; mkdir -p $O
for ord in lattice shuffled; do for W in 3 6 15 30; do
  UVOL_SIMT_W=$W timeout 600 python bench.py --no-cpu-baseline --no-variants --steps 2 --warmup 1 --only geo --mesh-order $ord > $O/geo_${ord}_W$W.json 2>> $O/err.log
done; done
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); g=d["kernel_groups_ms_per_step"]
    print(sys.argv[1].split('/')[-1], round(d["value"],1), "fps", round(d["ms_per_step"],1), "ms |", " ".join("%s=%.1f"%(k.split('.')[1],v) for k,v in sorted(g.items(), key=lambda x:-x[1])[:4]))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
