#!/usr/bin/env python3
"""tools/uastc_timing.py [segments]: UASTC mode on one MI355X, 2048^2 x 5 segments resident in HBM: encode (frames/s with the 21 MB
per segment copied to the host, kernel time and its HBM rate) and the RGBA / ASTC transcodes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "universal-volumetric_amd"))
import numpy as np, torch, synth, uvol
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B, S = 5, 2048
tex = synth.texture_sequence(B, size=S, seed=0)
dev = [torch.from_numpy(a).cuda() for a in tex]
ptrs = []
keep = []
for s in range(nseg):
    seg = dev if s == 0 else [torch.roll(t, shifts=(4 * s) % S, dims=1).contiguous() for t in dev]
    keep.append(seg); ptrs += [t.data_ptr() for t in seg]
torch.cuda.synchronize()
cd = uvol.Codec(device=0, uastc=1)
out = cd.encode_texture_segments_dev(ptrs, B, S, S)
cd.profile(True); cd.profile_reset()
t = time.perf_counter(); out = cd.encode_texture_segments_dev(ptrs, B, S, S); dt = time.perf_counter() - t
rep = {g["name"]: g for g in cd.profile_report()}
k = rep["tex.uastc_encode"]
res = {"segments": nseg, "encode_frames_per_s_incl_d2h": nseg * B / dt, "encode_kernel_ms": k["total_ms"], "encode_kernel_frames_per_s": nseg * B / (k["total_ms"] * 1e-3),
       "encode_kernel_GBps": k["algo_bytes"] / (k["total_ms"] * 1e-3) / 1e9, "bytes_per_segment": len(out[0])}
dec = cd.decode_texture_segments(out[:2])
src = np.stack([np.asarray(a)[::-1] for a in tex]).astype(np.float64)
mse = float(np.mean((src[..., :3] - dec[0][..., :3].astype(np.float64)) ** 2))
res["psnr_rgb_db"] = 10 * float(np.log10(255.0 ** 2 / mse)); res["bits_per_texel"] = 8.0 * len(out[0]) / (B * S * S)
cd.profile_reset()
t = time.perf_counter(); a = cd.transcode_texture_segments_astc(out); dt = time.perf_counter() - t
rep = {g["name"]: g for g in cd.profile_report()}
k = rep["texdec.uastc_astc"]
res.update({"astc_frames_per_s_incl_copies": nseg * B / dt, "astc_kernel_ms": k["total_ms"], "astc_kernel_GBps": k["algo_bytes"] / (k["total_ms"] * 1e-3) / 1e9})
print(json.dumps(res))
