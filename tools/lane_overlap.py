#!/usr/bin/env python3
"""tools/lane_overlap.py <kernel_trace.csv>: how much the groups of a pass overlap on the context's lanes.  From a rocprofv3 --kernel-trace CSV of
`bench.py --only geo`: for every kernel its class (serial walker kernels: walk / traversals / entropy / valence; front end and other
streaming kernels), then over the span of the trace the time during which (a) some serial kernel runs, (b) some streaming kernel runs,
(c) both at once, (d) two serial kernels of different streams at once.  One JSON line."""
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
SER = ("k_eb_walk", "k_traverse", "k_entropy", "k_eb_valence", "k_eb_ctx")
ev = []
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if not n.startswith("k_"): continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ser = n.startswith(SER)
    ev.append((s, 1, ser, r.get("Stream_Id") or r.get("Queue_Id"))); ev.append((e, -1, ser, r.get("Stream_Id") or r.get("Queue_Id")))
ev.sort()
ns = {True: 0, False: 0}; t_prev = ev[0][0]; acc = dict(serial=0, streaming=0, both=0, idle=0, serial_x2=0)
ser_streams = {}
for t, d, ser, st in ev:
    dt = t - t_prev
    if dt > 0:
        if ns[True]: acc["serial"] += dt
        if ns[False]: acc["streaming"] += dt
        if ns[True] and ns[False]: acc["both"] += dt
        if not ns[True] and not ns[False]: acc["idle"] += dt
        if sum(1 for v in ser_streams.values() if v > 0) >= 2: acc["serial_x2"] += dt
    ns[ser] += d
    if ser: ser_streams[st] = ser_streams.get(st, 0) + d
    t_prev = t
span = ev[-1][0] - ev[0][0]
print(json.dumps({"span_ms": span / 1e6, **{k + "_ms": v / 1e6 for k, v in acc.items()}, "streaming_hidden_behind_serial_frac": acc["both"] / max(1, acc["streaming"]),
                  "kernels": len(ev) // 2}))
