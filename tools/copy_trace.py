#!/usr/bin/env python3
"""tools/copy_trace.py <dir>: summary of a rocprofv3 --kernel-trace --memory-copy-trace run of a host-input bench: host -> device copies by
size class (count, bytes, busy time, rate while busy), the union of their busy intervals against the span (how long the link idled), the
largest idle gaps, and any runtime blit kernels among the kernels.  Diagnostic; prints JSON."""
import csv, glob, json, sys, re
d = sys.argv[1]
mc = sorted(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True))
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
out = {}
if mc:
    rows = list(csv.DictReader(open(mc[-1])))
    out["columns"] = list(rows[0].keys()) if rows else []
    h2d = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        direction = r.get("Direction", r.get("Kind", ""))
        nb = int(r.get("Size", r.get("Bytes", 0)) or 0)
        if "HOST_TO_DEVICE" in direction.upper() or "H2D" in direction.upper():
            h2d.append((s, e, nb))
    out["h2d_copies"] = len(h2d)
    if h2d:
        h2d.sort()
        # keep the second half of the run (the timed steps)
        t_lo = h2d[len(h2d) // 2][0]
        sel = [x for x in h2d if x[0] >= t_lo]
        span = max(e for s, e, n in sel) - sel[0][0]
        tot = sum(n for s, e, n in sel)
        merged = []
        for s, e, n in sel:
            if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
            else: merged.append([s, e])
        busy = sum(e - s for s, e in merged)
        gaps = sorted(((merged[i + 1][0] - merged[i][1]) / 1e6 for i in range(len(merged) - 1)), reverse=True)[:12]
        classes = {}
        for s, e, n in sel:
            k = "<1MiB" if n < (1 << 20) else "<16MiB" if n < (16 << 20) else "<128MiB" if n < (128 << 20) else ">=128MiB"
            c = classes.setdefault(k, [0, 0, 0]); c[0] += 1; c[1] += n; c[2] += e - s
        out["second_half"] = {"span_ms": span / 1e6, "bytes_GB": tot / 1e9, "GBps_over_span": tot / span, "link_busy_ms": busy / 1e6, "GBps_while_busy": tot / busy,
                              "largest_idle_gaps_ms": [round(g, 1) for g in gaps],
                              "by_size": {k: {"copies": c[0], "GB": round(c[1] / 1e9, 2), "sum_ms": round(c[2] / 1e6, 1), "GBps_each": round(c[1] / max(1, c[2]), 1)} for k, c in classes.items()}}
if kt:
    names = {}
    for r in csv.DictReader(open(kt[-1])):
        n = re.sub(r"\(.*", "", r["Kernel_Name"])
        if "rocclr" in n or "copyBuffer" in n or "fillBuffer" in n or "Blit" in n:
            a = names.setdefault(n, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out["runtime_blit_kernels"] = {k: {"launches": v[0], "ms": round(v[1] / 1e6, 1)} for k, v in names.items()}
print(json.dumps(out, indent=1))
