# why does the default step run in two regimes (2850 - 3050 and 3250 - 3400 frames/s between identical processes)?  host-side phases of every geometry group (UVOL_TIMING=1), five processes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${XTAG:-r05_x}; rm -rf $O; mkdir -p $O
nproc > $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>/dev/null; lscpu | grep -i "numa\|socket\|model name" >> $O/host.txt
for i in 1 2 3 4 5; do
  UVOL_TIMING=1 timeout 900 python bench.py --no-variants --no-cpu-baseline --warmup 1 --parity-frames 0 > $O/run$i.json 2> $O/run$i.err
  python - <<PY
import json, re
d = json.load(open("$O/run$i.json"))
rows = [list(map(float, re.findall(r"host prepared ([\d.]+) ms, enqueued ([\d.]+), gpu done ([\d.]+), packed d2h ([\d.]+), copied out ([\d.]+) \(enter at ([\d.]+)\)", l)[0])) for l in open("$O/run$i.err") if "geo group n=1280" in l]
rows = rows[2:]          # (warm-up pass)
if rows:
    m = [sum(r[k] for r in rows) / len(rows) for k in range(5)]
    ent = [r[5] for r in rows]; gaps = [round(b - a) for a, b in zip(ent, ent[1:])]
    print("run$i", round(d["value"]), "fps", round(d["ms_per_step"]), "ms/step | per group: prepared %.0f enqueued %.0f gpu done %.0f d2h %.0f copied %.0f ms | submit-to-submit gaps" % tuple(m), gaps)
else: print("run$i", round(d["value"]), "no timing lines")
PY
  grep "geo group" $O/run$i.err | head -20 > $O/run$i.timing.txt; rm -f $O/run$i.err
done
