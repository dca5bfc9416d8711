#!/usr/bin/env python3
"""Geometry decode-path timing on a real GPU: N x (100k-vertex / 200k-face .drc of this codec) decoded in one batch.
usage: tools/gdec_timing.py [n_frames]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "universal-volumetric_amd"))
import numpy as np
import uvol, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
c = uvol.Codec(device=0, max_batch=n)
files = c.encode_mesh_batch([synth.sphere_mesh(frame=k) for k in range(4)])
files = [files[i % 4] for i in range(n)]
c.decode_mesh_batch(files, fetch=False)            # warm-up: allocates the workspaces of the whole batch
c.profile(True); c.profile_reset()
t = time.time(); res = c.decode_mesh_batch(files, fetch=False); dt = time.time() - t
c.profile(False)
c.decode_mesh_batch(files, views=True)            # allocates (and touches) the host arrays of the whole batch
t = time.time(); res2 = c.decode_mesh_batch(files, views=True); dt2 = time.time() - t      # arrays re-used: what a host that keeps its buffers sees
ar = uvol.PinnedArena(c.decode_arena_bytes(files))               # ... and what one sees whose arrays lie in uvol_host_alloc memory (DMA writes them where they are)
c.decode_mesh_batch(files, views=True, arena=ar)
t = time.time(); res3 = c.decode_mesh_batch(files, views=True, arena=ar); dt3 = time.time() - t
print(json.dumps(dict(frames=n, frames_per_s_with_fetch_to_pinned_host_arrays=n / dt3, drc_bytes=len(files[0]), wall_s=dt, frames_per_s=n / dt, frames_per_s_with_fetch_to_host=n / dt2, groups={g["name"]: round(g["total_ms"], 1) for g in c.profile_report()})))
