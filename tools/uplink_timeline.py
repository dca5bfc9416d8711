#!/usr/bin/env python3
"""tools/uplink_timeline.py <dir>: from a rocprofv3 --kernel-trace run of a host-input bench, the uplink's copy kernels (start, end, queue)
and the groups' chains (k_job_clear ... k_gather per queue; k_tex_skip ... k_tex_pack), relative to the first copy of the last third.  Diagnostic."""
import csv, glob, sys, re
d = sys.argv[1]
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(kt)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r["Queue_Id"]))
rows.sort()
cp = [r for r in rows if r[2].startswith("k_uplink_copy")]
if not cp:
    print("no k_uplink_copy"); sys.exit(0)
t0 = cp[len(cp) * 2 // 3][0]
ev = []
for s, e, n, q in rows:
    if s < t0: continue
    if n.startswith("k_uplink_copy"): ev.append((s, e, "COPY", q))
    elif n.startswith("k_job_clear"): ev.append((s, e, "geo_begin", q))
    elif n.startswith("k_gather"): ev.append((s, e, "geo_end", q))
    elif n.startswith("k_eb_walk"): ev.append((s, e, "walk", q))
    elif n.startswith("k_traverse"): ev.append((s, e, "traverse", q))
    elif n.startswith("k_tex_skip"): ev.append((s, e, "tex_begin", q))
    elif n == "k_tex_pack": ev.append((s, e, "tex_end", q))
for s, e, n, q in ev:
    print("%9.1f -> %9.1f (%7.1f ms) q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
