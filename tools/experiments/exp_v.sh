O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --only geo --mesh-order shuffled --steps 2 --warmup 1 --no-cpu-baseline > $O/geo_shuffled.json 2>> $O/err.log
cp $(find $O/kt -name bench_kernel_stats.csv | head -1) $O/geo_shuffled_kernel_stats.csv; rm -rf $O/kt
python - $O <<'P'
import json, csv, sys
o = sys.argv[1]
for l in open('%s/geo_shuffled.json' % o):
    if l.startswith('{'):
        d = json.loads(l); print('geo shuffled', round(d['value']), round(d['ms_per_step']), {k: round(v) for k, v in d['kernel_groups_ms_per_step'].items()})
rows = sorted(csv.DictReader(open(o + '/geo_shuffled_kernel_stats.csv')), key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:30]: print('%8.1f ms/step x%-3d %s' % (float(r['TotalDurationNs']) / 3e6, int(r['Calls']), r['Name'][:50]))
P
