#!/usr/bin/env python3
"""tools/pmc_merge.py <dir> [<dir> ...]: pmc_{geo,tex}_{FETCH_SIZE,WRITE_SIZE}.json (tools/pmc_pack.sh; later directories fill in
passes the first lacks) -> <dir>/pmc_all_kernels.json (every kernel, MB per frame, sorted) and <dir>/pmc_traffic.json (sums per
bench kernel group, what bench.py reports as roofline.traffic).  FETCH_SIZE on gfx950 counts 64 bytes per 128-byte request: tools/pmc_cal.sh
(profiles/r03_pmc_calibration.json) measured FETCH_SIZE = 0.500 x the bytes of streaming reads (4 and 16 bytes per lane) and 64 bytes per
random 8-byte gather; WRITE_SIZE is exact for streaming writes and counts a 32-byte sector per scattered 4-byte store.  Tables carry the
raw counters and the corrected traffic = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: "double it before comparing with a byte count")."""
import json, os, re, sys
dirs, FR = sys.argv[1:], 2160
GROUPS = {"geo.k4_eb_walk": ["k_eb_walk"], "geo.k5_traverse": ["k_traverse"], "geo.k4_eb_valence": ["k_eb_valence"], "geo.k7_entropy_encode": ["k_entropy"],
          "geo.k2_dedup": ["k_dedup", "k_dd_", "k_faces", "k_compact_faces", "k_coherence", "k_relabel", "k_ms_", "k_face_cidx"], "geo.k3_corner_table": ["k_he_", "k_hp_", "k_edge_match", "k_vert0"],
          "geo.k4b_renumber_seams": ["k_renumber", "k_seams", "k_seam_bits", "k_aseg"],
          "tex.k12_sel_tokens": ["k_sel_tokens"], "tex.k9_endpoint_fit": ["k_tex_fit"], "tex.k10_selector_codebook": ["k_sel_stats", "k_sel_assign", "k_sel_centroids", "k_sel_used", "k_vq_apply<16>", "k_vq_decide<16>", "k_vq_zero<16>", "k_copy_skipped"]}
def load(h, c):
    for d in dirs:
        f = "%s/pmc_%s_%s.json" % (d, h, c)
        if os.path.exists(f): return json.load(open(f))
    return None
rows, tot, missing = [], {}, []
for h, name in (("geo", "geometry"), ("tex", "texture")):
    F, W = load(h, "FETCH_SIZE"), load(h, "WRITE_SIZE")
    if F is None or W is None: missing.append(name); continue
    for k in sorted(set(F) | set(W)):
        if k.startswith(("at::", "__amd_rocclr")): continue           # torch kernels / runtime copies that build the resident inputs
        f, w = F.get(k, [0, 0.0]), W.get(k, [0, 0.0])
        rows.append({"kernel": k, "half": name, "launches": max(f[0], w[0]), "fetch_bytes": f[1], "write_bytes": w[1], "mb_per_frame": (f[1] + w[1]) / FR / 1e6, "mb_per_frame_corrected": (2 * f[1] + w[1]) / FR / 1e6})
    tot[name + "_mb_per_frame"] = sum(r["mb_per_frame"] for r in rows if r["half"] == name)
    tot[name + "_mb_per_frame_corrected"] = sum(r["mb_per_frame_corrected"] for r in rows if r["half"] == name)
rows.sort(key=lambda r: -r["mb_per_frame"])
o = dirs[0]
json.dump(dict(frames=FR, total_mb_per_frame=sum(v for k, v in tot.items() if not k.endswith("_corrected")), total_mb_per_frame_corrected=sum(v for k, v in tot.items() if k.endswith("_corrected")),
               correction="reads x 2 (FETCH_SIZE counts 64 of every 128 bytes on gfx950, profiles/r03_pmc_calibration.json), writes as counted",
               missing_halves=missing, kernels=rows, **tot), open(o + "/pmc_all_kernels.json", "w"), indent=1)
tr = {"frames_per_launch": FR, "frames_per_texture_launch": FR, "unit": "bytes", "kernels": {}}
for g, pat in GROUPS.items():
    sel = [r for r in rows if any(p in r["kernel"] for p in pat)]
    if not sel: continue
    fb, wb = sum(r["fetch_bytes"] for r in sel), sum(r["write_bytes"] for r in sel)
    tr["kernels"][g] = {"kernels": pat, "fetch_bytes_per_launch_raw": fb, "fetch_bytes_per_launch_x2": 2 * fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch_raw": fb + wb, "hbm_bytes_per_frame_raw": (fb + wb) / FR, "hbm_bytes_per_launch": 2 * fb + wb, "hbm_bytes_per_frame": (2 * fb + wb) / FR}
json.dump(tr, open(o + "/pmc_traffic.json", "w"), indent=1)
print("totals MB/frame %s missing %s" % ({k: round(v, 1) for k, v in tot.items()}, missing))
for r in rows[:24]: print("%-40s %-8s x%-4d fetch %7.1f write %7.1f MB/frame" % (r["kernel"][:40], r["half"], r["launches"], r["fetch_bytes"] / FR / 1e6, r["write_bytes"] / FR / 1e6))
