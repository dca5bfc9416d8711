O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --only tex --steps 2 --warmup 1 --no-cpu-baseline > $O/tex.json 2>> $O/err.log
cp $(find $O/kt -name bench_kernel_stats.csv | head -1) $O/tex_kernel_stats.csv; rm -rf $O/kt
python - $O <<'P'
import json, csv, sys
o = sys.argv[1]
for l in open('%s/tex.json' % o):
    if l.startswith('{'):
        d = json.loads(l); print('tex', round(d['value']), round(d['ms_per_step']), {k: round(v) for k, v in d['kernel_groups_ms_per_step'].items() if k.startswith('tex')})
rows = sorted(csv.DictReader(open(o + '/tex_kernel_stats.csv')), key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:30]: print('%8.1f ms/step x%-4d %s' % (float(r['TotalDurationNs']) / 3e6, int(r['Calls']), r['Name'][:60]))
P
