#!/usr/bin/env python3
"""tools/ingest_timing.py [n]: the two device-side ingest stages alone on one MI355X - n 2048^2 RGBA images as inflated PNG scanlines
through uvol_unfilter_png_batch_dev, n 100 k-vertex OBJ texts through uvol_parse_obj_batch_dev - with the per-stage HIP-event times."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "universal-volumetric_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, synth, uvol
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
cd = uvol.Codec(device=0)
rng = np.random.default_rng(0)
img = synth.texture_sequence(1, size=2048, seed=0)[0]
# filtered scanlines with a fixed filter per row (vectorised over the image: Paeth / Sub / Up / Average / None in turn)
from test_hipemu_tex import png_scanlines
raw = png_scanlines(img, rng)
m = synth.sphere_mesh(frame=0, seed=0)
ip, iu, inn = (m[k].reshape(-1, 3).astype(np.int64) + 1 for k in ("idx_pos", "idx_uv", "idx_nrm"))
txt = ("\n".join("v %.6f %.6f %.6f" % tuple(v) for v in m["pos"].tolist()) + "\n" + "\n".join("vt %.7f %.7f" % tuple(v) for v in m["uv"].tolist()) + "\n" +
       "\n".join("vn %.6f %.6f %.6f" % tuple(v) for v in m["nrm"].tolist()) + "\n" +
       "\n".join("f %d/%d/%d %d/%d/%d %d/%d/%d" % tuple(r) for r in np.stack([ip, iu, inn], -1).reshape(-1, 9).tolist()) + "\n").encode()
res = {"n": n, "png_raw_bytes": len(raw), "obj_text_bytes": len(txt)}
for name, fn in (("png", lambda: cd.unfilter_png_batch_dev([raw] * n, 2048, 2048, 4, slot=0)), ("obj", lambda: cd.parse_obj_batch_dev([txt] * n, slot=0))):
    fn()
    cd.profile(True); cd.profile_reset()
    t = time.perf_counter(); fn(); dt = time.perf_counter() - t
    res[name] = {"wall_ms": 1e3 * dt, "per_s": n / dt, "groups_ms": {g["name"]: round(g["total_ms"], 2) for g in cd.profile_report()}}
    cd.profile(False)
print(json.dumps(res))
