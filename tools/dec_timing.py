#!/usr/bin/env python3
"""Decode-path timing on a real GPU: N segments of 5 x 2048^2 layers (this codec's own .ktx2 output), batched decode into
device buffers; prints frames/s and the per-kernel-group times.  usage: tools/dec_timing.py [n_segments] [size]"""
import json, os, sys, time
import torch
torch.zeros(1, device="cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "universal-volumetric_amd"))
import numpy as np
import uvol, synth
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 48
size = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
c = uvol.Codec(device=0)
tex = synth.texture_sequence(5, size=size, seed=3)
seg = c.encode_texture_segment(tex)
files = [seg] * nseg
bufs = torch.empty((nseg, 5, size, size, 4), dtype=torch.uint8, device="cuda:0")
ptrs = [bufs[s, l].data_ptr() for s in range(nseg) for l in range(5)]
c.decode_texture_segments_dev(files[:2], ptrs[:10], size * size * 4)          # warm-up (allocations)
c.profile(True); c.profile_reset()
torch.cuda.synchronize(); t = time.time()
c.decode_texture_segments_dev(files, ptrs, size * size * 4)
torch.cuda.synchronize(); dt = time.time() - t
rep = c.profile_report()
c.profile(False)
nh = min(nseg, 48)                                                      # host outputs (20 MB per layer): arrays kept between calls
host = c.decode_texture_segments(files[:nh])
t = time.time(); c.decode_texture_segments(files[:nh], out=host); dth = time.time() - t
print(json.dumps(dict(segments=nseg, layers=5, size=size, ktx2_bytes=len(seg), wall_s=dt, frames_per_s=nseg * 5 / dt,
                      rgba_GBps=nseg * 5 * size * size * 4 / dt / 1e9, frames_per_s_with_fetch_to_host=nh * 5 / dth, fetch_segments=nh, groups={g["name"]: round(g["total_ms"], 2) for g in rep})))
