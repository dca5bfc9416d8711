cd $GRAFT_REPO_ROOT; O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_geom.py -x -q -m gpu > $O/pytest_geom.log 2>&1
for F in 150 300; do
  UVOL_ENTROPY_WAVE=1 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --frames-per-step $F --only geo > $O/geo_${F}_entwave.json 2>> $O/err.log
  UVOL_WALK_PF=0 UVOL_ENTROPY_WAVE=1 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --frames-per-step $F --only geo > $O/geo_${F}_entwave_nopf.json 2>> $O/err.log
done
tail -3 $O/pytest_geom.log; tail -5 $O/err.log
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); g=d["kernel_groups_ms_per_step"]
    print(sys.argv[1].split('/')[-1], round(d["value"],1), "fps", round(d["ms_per_step"],1), "ms |", " ".join("%s=%.1f"%(k.split('.')[1],v) for k,v in sorted(g.items(), key=lambda x:-x[1])[:9]))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
