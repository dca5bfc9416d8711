#!/usr/bin/env python3
"""Runs the product's kernels - compiled for the host through tests/hipemu and with AddressSanitizer (`make -C universal-volumetric_amd
hipemu-asan`) - over the encode / decode paths and over corrupted decoder inputs.  The emulated device memory is the process heap, so an
out-of-bounds access of a KERNEL is an ASan report.  Test infrastructure (like tests/hipemu itself), not part of the product.

usage: LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_check.py [n_corruptions]
       (tools/asan_check.sh runs it under the kernel-form switches the tests use)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("oracle", "universal-volumetric_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
import oracle as o, synth, uvol
from test_hipemu_tex import _alpha_sequence

lib = os.environ.get("UVOL_SAN_LIB", os.path.join(ROOT, "tests", "hipemu", "libuvolcodec_hipemu_asan.so"))
ncorrupt = int(sys.argv[1]) if len(sys.argv) > 1 else 60
o.lib()
cd = uvol.Codec(lib_path=lib)
base = synth.sphere_mesh(40, 21, charts=(5, 4))
ms = [synth.torus_mesh(), base, synth.grid_mesh(), synth.shuffle_mesh(base, seed=3)] + list(synth.edge_case_meshes().values()) + [synth.random_soup_mesh(5)]
got = cd.encode_mesh_batch(ms)
for f, g in zip(ms, got):
    assert g == o.drc_encode(f["pos"], f["idx_pos"], f.get("uv"), f.get("idx_uv"), f.get("nrm"), f.get("idx_nrm"))
cd.decode_mesh_batch(got[:4])
files = [got[1]] + [o.drc_encode(base["pos"], base["idx_pos"], base["uv"], base["idx_uv"], base["nrm"], base["idx_nrm"], method=m) for m in (1, 2)]
cd.decode_mesh_batch(files)
tex = synth.texture_sequence(2, size=40, seed=1)
k1 = cd.encode_texture_segment(tex); assert k1 == o.ktx2_encode(tex)
ta = _alpha_sequence(2, 36, 2)
k2 = cd.encode_texture_segment(ta); assert k2 == o.ktx2_encode(ta)
cd.decode_texture_segments([k1]); cd.decode_texture_segments([k2]); cd.transcode_texture_segments_etc1([k1]); cd.transcode_texture_segments_bc7([k1])
for t in ("bc1", "bc3", "etc2_rgba", "bc7"):                    # every block target through the per-segment entry point, opaque and alpha files in one batch
    cd.transcode_texture_segments_status([k1, k2], t, shape=(40, 40, 2)); cd.transcode_texture_segments_status([k2], t, shape=(36, 36, 2))
c0 = uvol.Codec(lib_path=lib, DRACO_COMPRESSION_LEVEL=0); g0 = c0.encode_mesh_batch(ms[:3]); c0.decode_mesh_batch(g0); c0.close()
cu = uvol.Codec(lib_path=lib, uastc=1); ku = cu.encode_texture_segment(ta); cu.decode_texture_segments([ku]); cu.close()
# corrupted decoder inputs: bit flips, truncations, overwritten words - a clean error or a decoded result, never an out-of-bounds access
rng = np.random.default_rng(5)
outcomes = {}
for kind, data, first in (("drc", files[0], 11), ("drc_std", files[1], 11), ("drc_seq", files[2], 11), ("ktx2", k1, 80), ("ktx2_alpha", k2, 80), ("uastc", ku, 80)):
    ok = bad = 0
    for it in range(ncorrupt):
        b = bytearray(data); mode = it % 3
        if mode == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(first, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif mode == 1:
            b = b[:int(rng.integers(first + 1, len(b)))]
        else:
            p = int(rng.integers(first, len(b) - 4)); b[p:p + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
        try:
            if kind.startswith("drc"):
                cd.decode_mesh_batch([bytes(b)])
            elif kind == "uastc":
                c = uvol.Codec(lib_path=lib, uastc=1); c.decode_texture_segments([bytes(b)]); c.close()
            else:
                cd.decode_texture_segments([bytes(b)])
            ok += 1
        except (uvol.UvolError, ValueError):
            bad += 1
    outcomes[kind] = (ok, bad)
# the device inflate: valid streams of every block type, then corrupted ones (bit flips, truncations, overwritten words, random bytes) - a
# status per image, never an out-of-bounds access of the stream, the window ring or the scanline buffer
import zlib
from test_hipemu_tex import png_scanlines, zlib_variants
h, w, c = 48, 40, 4
img = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2)[..., None] + rng.integers(0, 6, (h, w, c))).astype(np.uint8)
raw = png_scanlines(img, rng); zs = zlib_variants(raw, rng)
ptrs, st = cd.inflate_png_batch_dev(zs, w, h, c); assert st == [0] * len(zs)
ok = bad = 0
for it in range(ncorrupt):
    b = bytearray(zs[it % len(zs)]); mode = it % 4
    if mode == 0:
        for _ in range(int(rng.integers(1, 4))): b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif mode == 1: b = b[:int(rng.integers(6, len(b)))]
    elif mode == 2: p_ = int(rng.integers(2, len(b) - 4)); b[p_:p_ + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
    else: b = bytearray(b[:2]) + bytearray(rng.integers(0, 256, int(rng.integers(8, 400)), dtype=np.uint8).tobytes())
    _, st = cd.inflate_png_batch_dev([zs[0], bytes(b), zs[1]], w, h, c, slot=it & 1)
    assert st[0] == 0 and st[2] == 0
    ok += st[1] == 0; bad += st[1] != 0
outcomes["inflate"] = (ok, bad)
# round 6: inputs in uvol_host_alloc memory travel through the context's uplink (slots on a copy stream, device layout mirroring the caller's
# arena, slots re-used behind release events, the texture call's deferred last part and its alpha re-run): enqueued calls in a row
ar = uvol.PinnedArena(64 << 20, lib_path=lib)
pm = [{k: ar.put(v) for k, v in m.items()} for m in ms[:6]]
want = got[:6]
assert cd.encode_mesh_batch(pm) == want
for _ in range(3): cd.start_mesh_batch(pm)
assert all(r == want for r in cd.finish())
segs = [synth.texture_sequence(2, size=40, seed=s_) for s_ in (1, 2, 3)] + [_alpha_sequence(2, 40, 5)]
wt = [o.ktx2_encode(s_) for s_ in segs]
ps = [[ar.put(a) for a in s_] for s_ in segs]
ct = uvol.Codec(lib_path=lib)
assert ct.encode_texture_segments(ps) == wt
for _ in range(3): ct.start_texture_segments(ps)
assert all(r == wt for r in ct.finish())
ct.trim(); assert ct.encode_texture_segments(ps[:2]) == wt[:2]
ct.close(); ar.close()
outcomes["uplink"] = "ok"
print("asan check passed:", outcomes)
