#!/usr/bin/env python3
"""tools/queue_timeline.py <dir> [t_from_ms t_to_ms]: per-queue kernel timeline of a rocprofv3 --kernel-trace run, consecutive launches of one
kernel merged, gaps of the queue > 3 ms shown; times relative to the first k_uplink_copy / first kernel of the last third.  Diagnostic."""
import csv, glob, sys, re
d = sys.argv[1]
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(kt)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    n = re.sub(r"<.*", "", n)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r["Queue_Id"]))
rows.sort()
t0 = rows[len(rows) * 2 // 3][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1500.0
byq = {}
for s, e, n, q in rows:
    ts, te = (s - t0) / 1e6, (e - t0) / 1e6
    if te < lo or ts > hi: continue
    byq.setdefault(q, []).append([ts, te, n, 1])
for q, rs in sorted(byq.items(), key=lambda kv: int(kv[0])):
    out = []
    for r in rs:
        if out and out[-1][2] == r[2] and r[0] - out[-1][1] < 1.0: out[-1][1] = max(out[-1][1], r[1]); out[-1][3] += 1
        else: out.append(list(r))
    print("== queue", q, "launches", len(rs))
    prev = None
    for ts, te, n, c in out:
        if prev is not None and ts - prev > 3.0: print("      ... idle %.1f ms" % (ts - prev))
        if te - ts > 2.0 or c > 50: print("   %8.1f -> %8.1f (%6.1f ms) x%-4d %s" % (ts, te, te - ts, c, n))
        prev = max(prev or te, te)
