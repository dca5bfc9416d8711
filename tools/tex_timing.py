#!/usr/bin/env python3
"""Quick per-kernel-group timing of the texture path on a real GPU (JSON to stdout)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "universal-volumetric_amd"))
import numpy as np
import uvol, synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cd = uvol.Codec(device=0)
tex = synth.texture_sequence(5, size=size, seed=0)
cd.encode_texture_segment(tex)
cd.profile(True); cd.profile_reset()
t = time.time(); k = cd.encode_texture_segment(tex); dt = time.time() - t
print(json.dumps(dict(size=size, wall_s=dt, segs_per_s=1 / dt, frames_per_s=5 / dt, bytes=len(k), groups=cd.profile_report()), indent=1))
