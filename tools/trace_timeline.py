#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV as a per-stream timeline of the last bench step (diagnostic)."""
import csv, sys, glob, re, os
THRESH = float(os.environ.get("TL_THRESH_MS", "0.8")) * 1e6
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/trace_r01/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
rows.sort(key=lambda r: r["s"])
# last step = from the last k_scan/k_dedup start ... find last 'k_minmax' or first geo kernel occurrence groups
walks = [r for r in rows if r["n"].startswith("k_eb_walk")]
last = walks[-1]
# step start: the last kernel gap > 5 ms before last walk?  use the first kernel after the previous step's k_gather
gathers = [r for r in rows if r["n"].startswith("k_gather") and r["e"] < last["s"]]
t0 = gathers[-1]["e"] if gathers else rows[0]["s"]
step = [r for r in rows if r["s"] >= t0]
T0 = step[0]["s"]
print("step kernels:", len(step), "span %.1f ms" % ((max(r["e"] for r in step) - T0) / 1e6))
bystream = {}
for r in step: bystream.setdefault((r["Queue_Id"], r["Stream_Id"]), []).append(r)
for k, rs in bystream.items():
    print("== queue/stream", k, "kernels", len(rs), "busy %.1f ms" % (sum(r["e"] - r["s"] for r in rs) / 1e6))
    # merge consecutive same-name kernels
    out = []; 
    for r in rs:
        if out and out[-1][0] == r["n"] and r["s"] - out[-1][2] < 2e6: out[-1][2] = r["e"]; out[-1][3] += 1
        else: out.append([r["n"], r["s"], r["e"], 1])
    for n, s, e, c in out:
        if (e - s) > THRESH or c > 20: print("   %8.1f -> %8.1f  (%7.1f ms) x%-4d %s" % ((s - T0) / 1e6, (e - T0) / 1e6, (e - s) / 1e6, c, n))
