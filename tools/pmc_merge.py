#!/usr/bin/env python3
"""tools/pmc_merge.py <dir>: merge pmc_all_kernels_{geo,tex}.json / pmc_traffic_{geo,tex}.json (tools/pmc_pack.sh) into
pmc_all_kernels.json / pmc_traffic.json (whichever halves exist)."""
import json, os, sys
o = sys.argv[1]
rows, tot, tr = [], {}, None
for h, name in (("geo", "geometry"), ("tex", "texture")):
    f = "%s/pmc_all_kernels_%s.json" % (o, h)
    if not os.path.exists(f): continue
    g = json.load(open(f))
    for r in g["kernels"]: r["half"] = name
    rows += g["kernels"]; tot[name + "_mb_per_frame"] = g["total_mb_per_frame"]
    t = json.load(open("%s/pmc_traffic_%s.json" % (o, h)))
    if tr is None: tr = t
    else: tr["kernels"].update(t["kernels"])
rows.sort(key=lambda r: -r["mb_per_frame"])
json.dump(dict(frames=2160, total_mb_per_frame=sum(tot.values()), kernels=rows, **tot), open(o + "/pmc_all_kernels.json", "w"), indent=1)
json.dump(tr, open(o + "/pmc_traffic.json", "w"), indent=1)
print("total %.1f MB/frame %s" % (sum(tot.values()), tot))
