#!/usr/bin/env python3
"""tools/latency.py: single-frame latency on one MI355X (one 100,002-vertex frame, one 2048^2 image, one call each, inputs on the host):
what a caller that cannot batch sees.  Median of 5 after a warm-up."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "universal-volumetric_amd"))
import numpy as np, synth, uvol
m = synth.sphere_mesh(frame=0, seed=0); tex = synth.texture_sequence(1, size=2048, seed=0)
res = {}
for name, cfg in (("etc1s", {}), ("uastc", {"uastc": 1})):
    cd = uvol.Codec(device=0, **cfg)
    cd.encode_mesh(**m); cd.encode_texture_segment(tex)
    tg, tt = [], []
    for _ in range(5):
        t = time.perf_counter(); d = cd.encode_mesh(**m); tg.append(time.perf_counter() - t)
        t = time.perf_counter(); k = cd.encode_texture_segment(tex); tt.append(time.perf_counter() - t)
    res[name] = {"mesh_ms": 1e3 * float(np.median(tg)), "texture_1_layer_ms": 1e3 * float(np.median(tt)), "drc_bytes": len(d), "ktx2_bytes": len(k)}
    cd.close()
print(json.dumps({"what": "single-frame latency, host buffers in, host bytes out, one call per frame", **res}))
