#!/usr/bin/env python3
"""Quick per-kernel-group timing of the geometry path on a real GPU (writes JSON to stdout)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "universal-volumetric_amd"))
import numpy as np
import uvol, synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cd = uvol.Codec(device=0, max_batch=nb)
frames = [synth.sphere_mesh(frame=k) for k in range(nb)]
cd.encode_mesh_batch(frames[:2])          # warm-up (allocations)
cd.profile(True); cd.profile_reset()
t = time.time(); res = cd.encode_mesh_batch(frames); dt = time.time() - t
rep = cd.profile_report()
print(json.dumps(dict(batch=nb, wall_s=dt, fps=nb / dt, bytes=[len(r) for r in res][:4], groups=rep), indent=1))
