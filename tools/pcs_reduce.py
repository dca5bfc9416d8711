#!/usr/bin/env python3
"""tools/pcs_reduce.py <rocprof_dir> <out.txt> <kernel substring>: rocprofv3 PC-sampling CSV -> samples per instruction of the kernels whose
name contains the substring (joined to the kernel trace by dispatch id), most sampled first; with stochastic sampling also the stall reasons."""
import collections, csv, glob, sys
d, outp, pat = sys.argv[1], sys.argv[2], sys.argv[3]
out = open(outp, "w")
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
disp = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        disp[r.get("Dispatch_Id")] = r.get("Kernel_Name", "")
for f in glob.glob(d + "/**/*pc_sampling*.csv", recursive=True):
    rd = csv.DictReader(open(f))
    print("FILE", f, rd.fieldnames, file=out)
    hist, extra, n, shown = collections.Counter(), collections.defaultdict(collections.Counter), 0, 0
    for r in rd:
        if shown < 3: print(dict(r), file=out); shown += 1
        k = disp.get(r.get("Dispatch_Id"), "")
        if pat not in k: continue
        n += 1
        key = (r.get("Instruction") or r.get("Instruction_Comment") or r.get("Pc_Offset") or "?")
        hist[key] += 1
        for c in ("Wave_Issued_Instruction", "Stall_Reason", "Instruction_Type", "Arb_State_Issue", "Arb_State_Stall", "Snapshot_Stall_Reason"):
            if c in r: extra[c][r[c]] += 1
    print("samples in", pat, n, file=out)
    for c, h in extra.items(): print(c, h.most_common(12), file=out)
    for k, v in hist.most_common(90): print("%7d %5.1f%%  %s" % (v, 100.0 * v / max(n, 1), k), file=out)
