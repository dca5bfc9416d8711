#!/usr/bin/env python3
"""tools/pmc_one.py <rocprof_dir> <COUNTER> <out.json>: per-kernel sums of one rocprofv3 --pmc pass ({kernel: [launches, bytes]};
counter unit KiB).  Kernel names without their argument lists; torch / runtime set-up kernels are kept (tools/pmc_merge.py drops them)."""
import collections, csv, glob, json, re, sys
d, c, outp = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/bench_counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == c:
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        a = agg[n]; a[0] += 1; a[1] += float(r["Counter_Value"]) * 1024
json.dump(agg, open(outp, "w"))
print(c, len(agg), "kernels", round(sum(v[1] for v in agg.values()) / 1e9, 1), "GB")
