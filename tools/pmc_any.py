#!/usr/bin/env python3
"""tools/pmc_any.py <rocprof_dir> <out.json> [kernel substring ...]: per-kernel sums of EVERY counter of one rocprofv3 --pmc pass
({kernel: {counter: sum, "launches": n}}), raw counter values; kernels filtered by substring when given (diagnostics of the walker kernels)."""
import collections, csv, glob, json, re, sys
d, outp, pats = sys.argv[1], sys.argv[2], sys.argv[3:]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if pats and not any(p in n for p in pats): continue
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"]); disp[n].add(r.get("Dispatch_Id", ""))
out = {n: dict(v, launches=len(disp[n])) for n, v in agg.items()}
json.dump(out, open(outp, "w"), indent=1)
for n, v in out.items(): print(n[:40], {k: round(x, 1) for k, x in v.items()})
