# round 4: decode path with the attribute symbol streams on the second stream; GPU decode tests; when the kernels ran (kernel trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4z; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "decode or resident or transcode" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python tools/gdec_timing.py 1920 > $O/gdec_timing.json 2> $O/gdec.err
GPU_MAX_HW_QUEUES=24 timeout 600 python tools/gdec_timing.py 1920 > $O/gdec_timing_q24.json 2>> $O/gdec.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o b -- python tools/gdec_timing.py 1920 > $O/gdec_timing_traced.json 2>> $O/gdec.err
python - <<'PY' > $O/trace_rows.txt 2>&1
import csv, glob, os
f = glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r4z/kt/**/b_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gdec" in r["Kernel_Name"] or "traverse" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows[-40:]:
    print("%-40s q%-3s %9.1f -> %9.1f ms" % (r["Kernel_Name"][:40], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
PY
rm -rf $O/kt
