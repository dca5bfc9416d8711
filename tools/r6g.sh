cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g; mkdir -p $O
for bgk in off h2d_16m h2d_256m d2d; do
  timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 --only geo --steps 4 --warmup 1 --background $bgk > $O/geo_$bgk.json 2>> $O/err
  python - $O/geo_$bgk.json $bgk <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("geo", sys.argv[2], round(d["value"]), "frames/s", d.get("background"), {k:round(v) for k,v in list(d["kernel_groups_ms_per_step"].items())[:4]})
PY
done
for bgk in off h2d_16m h2d_256m; do
  timeout 600 python bench.py --no-cpu-baseline --no-variants --parity-frames 0 --only tex --steps 4 --warmup 1 --background $bgk > $O/tex_$bgk.json 2>> $O/err
  python - $O/tex_$bgk.json $bgk <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("tex", sys.argv[2], round(d["value"]), "frames/s", d.get("background"), {k:round(v) for k,v in list(d["kernel_groups_ms_per_step"].items())[:4]})
PY
done
