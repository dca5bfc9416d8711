#!/usr/bin/env python3
"""tools/inflate_timing.py [n ...]: the device-side PNG inflate alone on one MI355X - n zlib streams of 2048^2 RGBA scanlines (four
images x zlib levels 1 / 6 / 9, cycled) through uvol_inflate_png_batch_dev, with the HIP-event times of the inflate and un-filter
kernels and, beside them, the host zlib on one core."""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "universal-volumetric_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, synth, uvol
from test_hipemu_tex import png_scanlines
ns = [int(a) for a in sys.argv[1:]] or [240, 960]
cd = uvol.Codec(device=0)
rng = np.random.default_rng(0)
raws = [png_scanlines(t, rng) for t in synth.texture_sequence(4, size=2048, seed=0)]
zs = [zlib.compress(r, lv) for r in raws for lv in (1, 6, 9)]
t = time.perf_counter(); [zlib.decompress(z) for z in zs]; host = (time.perf_counter() - t) / len(zs)
res = {"what": "uvol_inflate_png_batch_dev alone: zlib streams of 2048^2 RGBA PNG scanlines (4 images x levels 1 / 6 / 9, cycled); k_inflate is one wave per stream",
       "stream_bytes": [len(z) for z in zs], "inflated_bytes": len(raws[0]), "host_zlib_ms_per_image_one_core": round(1e3 * host, 1), "rows": {}}
for n in ns:
    batch = [zs[i % len(zs)] for i in range(n)]
    cd.inflate_png_batch_dev(batch, 2048, 2048, 4, slot=0)
    cd.profile(True); cd.profile_reset()
    t = time.perf_counter(); _, st = cd.inflate_png_batch_dev(batch, 2048, 2048, 4, slot=0); dt = time.perf_counter() - t
    assert not any(st)
    g = {x["name"]: round(x["total_ms"], 2) for x in cd.profile_report()}
    cd.profile(False)
    res["rows"][str(n)] = {"wall_ms": round(1e3 * dt, 1), "images_per_s_wall": round(n / dt, 1), "groups_ms": g,
                           "images_per_s_inflate_kernel": round(n / (g.get("ingest.png_inflate", 0) / 1e3), 1) if g.get("ingest.png_inflate") else None}
print(json.dumps(res))
