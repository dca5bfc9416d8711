# dedup in frame groups: group size sweep (geometry only + the full path)
O=gpurun_out/r02_o; mkdir -p $O
for g in 2160 128 32 8; do UVOL_DD_GROUP=$g timeout 300 python bench.py --only geo --steps 2 --warmup 1 --no-cpu-baseline > $O/geo_g$g.json 2>> $O/err.log; done
for g in 2160 32; do UVOL_DD_GROUP=$g timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/full_g$g.json 2>> $O/err.log; done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02_o/*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l); g = d['kernel_groups_ms_per_step']
            print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']), {k: round(v) for k, v in g.items() if k.startswith('geo')})
P
