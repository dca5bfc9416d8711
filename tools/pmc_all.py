#!/usr/bin/env python3
"""tools/pmc_all.py <fetch_dir> <write_dir> <frames_per_step> <out.json>: HBM bytes per step and per frame of EVERY kernel from the
FETCH_SIZE / WRITE_SIZE passes (counter unit KiB; raw FETCH_SIZE, see tools/pmc_summary.py), sorted by total traffic."""
import collections, csv, json, re, sys
fd, wd, frames, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
def load(d, c):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"{d}/bench_counter_collection.csv")):
        if r["Counter_Name"] == c:
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            a = agg[n]; a[0] += 1; a[1] += float(r["Counter_Value"]) * 1024
    return agg
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
rows = []
for k in sorted(set(F) | set(W)):
    f, w = F.get(k, [0, 0.0]), W.get(k, [0, 0.0])
    rows.append({"kernel": k, "launches": max(f[0], w[0]), "fetch_bytes": f[1], "write_bytes": w[1], "mb_per_frame": (f[1] + w[1]) / frames / 1e6})
rows.sort(key=lambda r: -(r["fetch_bytes"] + r["write_bytes"]))
setup = [r for r in rows if r["kernel"].startswith(("at::", "__amd_rocclr"))]       # torch kernels / runtime copies that build the resident inputs, not the path
rows = [r for r in rows if r not in setup]
tot = sum(r["fetch_bytes"] + r["write_bytes"] for r in rows)
json.dump({"frames": frames, "excluded_setup_kernels_mb_per_frame": sum(r["mb_per_frame"] for r in setup), "total_bytes": tot, "total_mb_per_frame": tot / frames / 1e6, "kernels": rows}, open(outp, "w"), indent=1)
print("total MB/frame %.1f" % (tot / frames / 1e6))
for r in rows[:30]: print("%-44s x%-5d fetch %8.1f MB/frame  write %8.1f MB/frame" % (r["kernel"][:44], r["launches"], r["fetch_bytes"] / frames / 1e6, r["write_bytes"] / frames / 1e6))
