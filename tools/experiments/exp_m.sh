# per-kernel check of an experiment: geometry-only kernel trace at 2160 frames + the full default line
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --only geo --steps 2 --warmup 1 --no-cpu-baseline > $O/geo.json 2>> $O/err.log
cp $(find $O/kt -name bench_kernel_stats.csv | head -1) $O/geo_kernel_stats.csv; rm -rf $O/kt
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/full.json 2>> $O/err.log
python - $O <<'P'
import json, csv, sys
o = sys.argv[1]
for f in ('geo', 'full'):
    for l in open('%s/%s.json' % (o, f)):
        if l.startswith('{'):
            d = json.loads(l); print(f, round(d['value']), round(d['ms_per_step']), {k: round(v) for k, v in d['kernel_groups_ms_per_step'].items() if k.startswith('geo')})
rows = sorted(csv.DictReader(open(o + '/geo_kernel_stats.csv')), key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:36]: print('%8.1f ms/step x%-3d %s' % (float(r['TotalDurationNs']) / 3e6, int(r['Calls']), r['Name'][:50]))
P
