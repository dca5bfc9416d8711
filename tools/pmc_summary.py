#!/usr/bin/env python3
"""tools/pmc_summary.py <fetch_dir> <write_dir> <frames_per_geometry_launch>[:<frames_per_texture_launch>] <out.json>
Summarise two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are collected in separate runs: they do not fit the 4 TCC
slots together, MI355X_MICROARCH.md §rocprofv3 PMC slots) into HBM bytes per launch / per frame for the main kernels.
Counter unit is KiB.  On gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read (guide §HBM); both the raw
and the doubled figure are recorded and the raw one is used for `traffic` (a lower bound for the read side)."""
import collections, csv, json, sys
fd, wd, outp = sys.argv[1], sys.argv[2], sys.argv[4]
frames = int(sys.argv[3].split(':')[0]); frames_tex = int(sys.argv[3].split(':')[1]) if ':' in sys.argv[3] else frames
GROUPS = {"geo.k4_eb_walk": ["k_eb_walk"], "geo.k5_traverse": ["k_traverse"], "geo.k4_eb_valence": ["k_eb_valence"], "geo.k7_entropy_encode": ["k_entropy"],
          "geo.k2_dedup": ["k_dedup", "k_dd_", "k_faces", "k_compact_faces"], "geo.k3_corner_table": ["k_he_", "k_hp_", "k_edge_match", "k_vert0"],
          "geo.k4b_renumber_seams": ["k_renumber", "k_seams", "k_seam_bits", "k_aseg"],
          "tex.k12_sel_tokens": ["k_sel_tokens"], "tex.k9_endpoint_fit": ["k_tex_fit"], "tex.k10_selector_codebook": ["k_sel_stats", "k_sel_assign", "k_sel_centroids", "k_sel_used", "k_vq_apply<16>", "k_vq_decide<16>", "k_vq_zero<16>", "k_copy_skipped"]}
def load(d, c):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"{d}/bench_counter_collection.csv")):
        if r["Counter_Name"] == c:
            a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
out = {"frames_per_launch": frames, "frames_per_texture_launch": frames_tex, "unit": "bytes", "kernels": {}}
for g, pat in GROUPS.items():
    f = [v for k, v in F.items() if any(p_ in k for p_ in pat)]; w = [v for k, v in W.items() if any(p_ in k for p_ in pat)]
    if not f or not w: continue
    # bytes per LAUNCH OF THE GROUP (one batch): the sum over the group's kernels of one step
    fb = sum(v[1] for v in f) * 1024; wb = sum(v[1] for v in w) * 1024
    out["kernels"][g] = {"kernels": pat, "fetch_bytes_per_launch_raw": fb, "fetch_bytes_per_launch_x2": 2 * fb, "write_bytes_per_launch": wb,
                         "hbm_bytes_per_launch": fb + wb, "hbm_bytes_per_frame": (fb + wb) / (frames_tex if g.startswith("tex.") else frames)}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_frame"] / 1e6, 2) for k, v in out["kernels"].items()}), "MB/frame")
