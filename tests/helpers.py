"""Shared test helpers: canonical face-multiset comparison (draco reorders vertices/faces)."""
import numpy as np


def quant(v, bits):
    v = np.asarray(v, dtype=np.float32)
    mn = v.min(0); rng = np.float32((v.max(0) - mn).max())
    if rng == 0:
        rng = np.float32(1)
    inv = np.float32((1 << bits) - 1) / rng
    return np.floor((v - mn) * inv + np.float32(0.5)).astype(np.int64), mn, rng


def facekeys(K):
    """K: (nf,3,k) int keys -> faces rotated so the lexicographically smallest corner is first, then row-sorted."""
    nf = K.shape[0]
    flat = K.reshape(nf, 3, -1)
    idx = np.zeros(nf, dtype=np.int64)
    ar = np.arange(nf)
    for j in (1, 2):
        a = flat[ar, idx]; b = flat[:, j]
        less = np.zeros(nf, bool); eq = np.ones(nf, bool)
        for c in range(flat.shape[2]):
            less |= eq & (b[:, c] < a[:, c]); eq &= (b[:, c] == a[:, c])
        idx = np.where(less, j, idx)
    r = np.stack([flat[ar, (idx + k) % 3] for k in range(3)], 1).reshape(nf, -1)
    return r[np.lexsort(r.T[::-1])]


def check_roundtrip(O, mesh, drc_bytes, qp=11, qt=10):
    """decode(drc) must reproduce the input: identical triangle multiset over (pos_q, uv_q) keys,
    positions within half a quantisation step (+1 ulp), zero leftover bytes."""
    d = O.drc_decode(drc_bytes)
    assert d.leftover == 0
    pos = np.asarray(mesh["pos"], np.float32).reshape(-1, 3)
    ip = np.asarray(mesh["idx_pos"]).reshape(-1)
    pq, pmn, prng = quant(pos, qp)
    keys_in = [pq[ip]]
    p = d.att("position")
    keys_out = [p["vals"][p["corner_to_entry"]].astype(np.int64)]
    if mesh.get("uv") is not None and d.att("tex_coord") is not None:
        uv = np.asarray(mesh["uv"], np.float32).reshape(-1, 2)
        uq, _, _ = quant(uv, qt)
        keys_in.append(uq[np.asarray(mesh["idx_uv"]).reshape(-1)])
        u = d.att("tex_coord")
        keys_out.append(u["vals"][u["corner_to_entry"]].astype(np.int64))
    Kin = np.concatenate(keys_in, 1); Kout = np.concatenate(keys_out, 1)
    # the encoder drops faces that are degenerate after value dedup
    F = Kin.reshape(-1, 3, Kin.shape[1])
    pid = pos[ip].reshape(-1, 3, 3)
    deg = (pid[:, 0] == pid[:, 1]).all(1) | (pid[:, 1] == pid[:, 2]).all(1) | (pid[:, 0] == pid[:, 2]).all(1)
    a = facekeys(F[~deg]); b = facekeys(Kout.reshape(-1, 3, Kout.shape[1]))
    assert a.shape == b.shape and np.array_equal(a, b), "triangle index arrays differ after canonicalisation"
    step = prng / np.float32((1 << qp) - 1)
    assert abs(p["range"] - prng) <= 1e-6 * max(1.0, abs(prng))
    # the decoded integers equal pq (checked above through the triangle multiset), so the decoder-side
    # reconstruction of every input vertex is minv + pq * range/(2^qp-1) with the header's minv/range:
    recon = np.asarray(p["minv"][:3], np.float32) + pq.astype(np.float32) * (np.float32(p["range"]) / np.float32((1 << qp) - 1))
    err = np.abs(recon - pos).max()
    assert err <= step / 2 * 1.001 + 1e-4 * max(1.0, float(np.abs(pos).max())), (err, step)
    return d


def etc1_decode_blocks(blocks, width, height):
    """Independent ETC1 block decoder (Khronos data-format spec, ETC1 / ETC2 RGB8 'individual' and 'differential' modes) used to
    check the ETC1 transcode target: blocks [by, bx, 8] uint8 -> RGBA8 [height, width, 4]."""
    MOD = np.array([[2, 8], [5, 17], [9, 29], [13, 42], [18, 60], [24, 80], [33, 106], [47, 183]], np.int32)
    by, bx = blocks.shape[:2]
    out = np.zeros((by * 4, bx * 4, 4), np.uint8); out[..., 3] = 255
    b = blocks.astype(np.int32)
    diff = (b[..., 3] >> 1) & 1; flip = b[..., 3] & 1
    t1 = (b[..., 3] >> 5) & 7; t2 = (b[..., 3] >> 2) & 7
    base1 = np.zeros((by, bx, 3), np.int32); base2 = np.zeros((by, bx, 3), np.int32)
    for c in range(3):
        hi5 = b[..., c] >> 3; d3 = b[..., c] & 7; d3 = np.where(d3 >= 4, d3 - 8, d3)
        c1d = (hi5 << 3) | (hi5 >> 2); s5 = hi5 + d3; c2d = (s5 << 3) | (s5 >> 2)
        hi4 = b[..., c] >> 4; lo4 = b[..., c] & 15
        base1[..., c] = np.where(diff == 1, c1d, hi4 * 17); base2[..., c] = np.where(diff == 1, c2d, lo4 * 17)
    msb = (b[..., 4] << 8) | b[..., 5]; lsb = (b[..., 6] << 8) | b[..., 7]
    for x in range(4):
        for y in range(4):
            i = 4 * x + y
            idx = (((msb >> i) & 1) << 1) | ((lsb >> i) & 1)
            second = np.where(flip == 1, y >= 2, x >= 2)
            tab = np.where(second, t2, t1)
            mag = MOD[tab, idx & 1]; mod = np.where(idx >= 2, -mag, mag)
            base = np.where(second[..., None], base2, base1)
            out[y::4, x::4, :3] = np.clip(base + mod[..., None], 0, 255).astype(np.uint8)
    return out[:height, :width]
