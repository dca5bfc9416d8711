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


def eac_alpha_decode_blocks(blocks, width, height):
    """Independent EAC alpha (ETC2 RGBA8's first 8 bytes) decoder from the public format description: byte 0 base codeword, byte 1
    multiplier << 4 | table, then 16 x 3-bit indices (pixel i = 4 * x + y, first pixel in the top bits); alpha = clamp(base +
    multiplier * table[index]).  blocks [by, bx, 8] uint8 -> alpha [height, width] uint8."""
    T = np.array([[-3, -6, -9, -15, 2, 5, 8, 14], [-3, -7, -10, -13, 2, 6, 9, 12], [-2, -5, -8, -13, 1, 4, 7, 12], [-2, -4, -6, -13, 1, 3, 5, 12],
                  [-3, -6, -8, -12, 2, 5, 7, 11], [-3, -7, -9, -11, 2, 6, 8, 10], [-4, -7, -8, -11, 3, 6, 7, 10], [-3, -5, -8, -11, 2, 4, 7, 10],
                  [-2, -6, -8, -10, 1, 5, 7, 9], [-2, -5, -8, -10, 1, 4, 7, 9], [-2, -4, -8, -10, 1, 3, 7, 9], [-2, -5, -7, -10, 1, 4, 6, 9],
                  [-3, -4, -7, -10, 2, 3, 6, 9], [-1, -2, -3, -10, 0, 1, 2, 9], [-4, -6, -8, -9, 3, 5, 7, 8], [-3, -5, -7, -9, 2, 4, 6, 8]], np.int32)
    by, bx = blocks.shape[:2]
    b = blocks.astype(np.int64)
    base = b[..., 0]; mult = b[..., 1] >> 4; tab = b[..., 1] & 15
    bits = np.zeros((by, bx), np.int64)
    for k in range(6):
        bits = (bits << 8) | b[..., 2 + k]
    out = np.zeros((by * 4, bx * 4), np.uint8)
    for x in range(4):
        for y in range(4):
            i = 4 * x + y
            idx = (bits >> (45 - 3 * i)) & 7
            out[y::4, x::4] = np.clip(base + mult * T[tab, idx], 0, 255).astype(np.uint8)
    return out[:height, :width]


def bc7_decode_blocks(blocks, width, height):
    """Independent BC7 decoder for the two single-subset modes the BC7 transcode target emits (Khronos data-format spec, BPTC):
    mode 5 (7-bit RGB endpoints, 8-bit alpha endpoints, 2-bit colour and alpha indices, any rotation) and mode 6 (7-bit RGBA
    endpoints + p-bits, 4-bit indices).  blocks [by, bx, 16] uint8 -> RGBA8 [height, width, 4]; other modes assert."""
    W2 = np.array([0, 21, 43, 64], np.int64)
    W4 = np.array([0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64], np.int64)
    by, bx = blocks.shape[:2]
    b = blocks.reshape(by, bx, 16).astype(np.uint64)
    lo = np.zeros((by, bx), np.uint64); hi = np.zeros((by, bx), np.uint64)
    for i in range(8):
        lo |= b[..., i] << np.uint64(8 * i); hi |= b[..., 8 + i] << np.uint64(8 * i)

    def bits(pos, n):
        mask = np.uint64((1 << n) - 1)
        if pos >= 64:
            return ((hi >> np.uint64(pos - 64)) & mask).astype(np.int64)
        v = lo >> np.uint64(pos)
        if pos + n > 64:
            v = v | (hi << np.uint64(64 - pos))
        return (v & mask).astype(np.int64)

    m5 = bits(0, 6) == 32; m6 = bits(0, 7) == 64
    assert np.all(m5 | m6), "block is neither BC7 mode 5 nor mode 6"
    out = np.zeros((by * 4, bx * 4, 4), np.uint8)
    # mode 6
    ep = np.zeros((by, bx, 4, 2), np.int64)
    for c in range(4):
        for k in range(2):
            ep[..., c, k] = (bits(7 + 14 * c + 7 * k, 7) << 1) | bits(63 + k, 1)
    pos = 65; px6 = []
    for i in range(16):
        n = 3 if i == 0 else 4
        w = W4[bits(pos, n)][..., None]; pos += n
        px6.append((ep[..., 0] * (64 - w) + ep[..., 1] * w + 32) >> 6)
    # mode 5
    rot = np.where(m5, bits(6, 2), 0)                        # 1 / 2 / 3: the scalar channel was R / G / B (swapped back after the interpolation)
    e5 = np.zeros((by, bx, 4, 2), np.int64)
    for c in range(3):
        for k in range(2):
            v = bits(8 + 14 * c + 7 * k, 7); e5[..., c, k] = (v << 1) | (v >> 6)
    e5[..., 3, 0] = bits(50, 8); e5[..., 3, 1] = bits(58, 8)
    cpos = 66; apos = 97; px5 = []
    for i in range(16):
        n = 1 if i == 0 else 2
        wc = W2[bits(cpos, n)][..., None]; wa = W2[bits(apos, n)]; cpos += n; apos += n
        rgb = (e5[..., :3, 0] * (64 - wc) + e5[..., :3, 1] * wc + 32) >> 6
        a = (e5[..., 3, 0] * (64 - wa) + e5[..., 3, 1] * wa + 32) >> 6
        v = np.concatenate([rgb, a[..., None]], axis=-1)
        for r in (1, 2, 3):
            sw = v.copy(); sw[..., r - 1] = v[..., 3]; sw[..., 3] = v[..., r - 1]
            v = np.where((rot == r)[..., None], sw, v)
        px5.append(v)
    for i in range(16):
        y, x = divmod(i, 4)
        out[y::4, x::4] = np.where(m5[..., None], px5[i], px6[i]).astype(np.uint8)
    return out[:height, :width]


def bc1_decode_blocks(blocks, width, height, four_colour_always=False):
    """Independent BC1 (DXT1) colour-block decoder written from the format description: colour0 / colour1 as little-endian RGB565, 16 2-bit
    indices in raster order from bit 0; colour0 > colour1 (or four_colour_always, as inside BC2 / BC3): palette c0, c1, (2 c0 + c1) / 3,
    (c0 + 2 c1) / 3; otherwise c0, c1, (c0 + c1) / 2 and transparent black.  blocks [by, bx, 8] uint8 -> RGBA8 [height, width, 4]."""
    by, bx = blocks.shape[:2]
    out = np.zeros((by * 4, bx * 4, 4), np.uint8)
    for y in range(by):
        for x in range(bx):
            b = [int(v) for v in blocks[y, x]]
            c0, c1 = b[0] | (b[1] << 8), b[2] | (b[3] << 8)
            def rgb(c):
                r, g, bl = c >> 11, (c >> 5) & 63, c & 31
                return np.array([(r << 3) | (r >> 2), (g << 2) | (g >> 4), (bl << 3) | (bl >> 2)], np.int64)
            e0, e1 = rgb(c0), rgb(c1)
            if c0 > c1 or four_colour_always:
                pal = [(e0, 255), (e1, 255), ((2 * e0 + e1) // 3, 255), ((e0 + 2 * e1) // 3, 255)]
            else:
                pal = [(e0, 255), (e1, 255), ((e0 + e1) // 2, 255), (np.zeros(3, np.int64), 0)]
            idx = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24)
            for i in range(16):
                c, a = pal[(idx >> (2 * i)) & 3]
                out[4 * y + i // 4, 4 * x + i % 4, :3] = c; out[4 * y + i // 4, 4 * x + i % 4, 3] = a
    return out[:height, :width]


def bc3_decode_blocks(blocks, width, height):
    """Independent BC3 (DXT5) decoder: a BC4 alpha block (alpha0, alpha1, 16 3-bit indices in raster order from bit 0; alpha0 > alpha1: six
    interpolated values ((8 - j) a0 + (j - 1) a1) / 7 for index j = 2 .. 7, else four interpolated values, then 0 and 255) followed by a
    BC1 colour block read in four-colour mode.  blocks [by, bx, 16] uint8 -> RGBA8 [height, width, 4]."""
    by, bx = blocks.shape[:2]
    out = bc1_decode_blocks(blocks[..., 8:], by * 4, bx * 4, four_colour_always=True).copy()
    for y in range(by):
        for x in range(bx):
            b = [int(v) for v in blocks[y, x, :8]]
            a0, a1 = b[0], b[1]
            if a0 > a1:
                pal = [a0, a1] + [((8 - j) * a0 + (j - 1) * a1) // 7 for j in range(2, 8)]
            else:
                pal = [a0, a1] + [((6 - j) * a0 + (j - 1) * a1) // 5 for j in range(2, 6)] + [0, 255]
            bits = sum(b[2 + k] << (8 * k) for k in range(6))
            for i in range(16):
                out[4 * y + i // 4, 4 * x + i % 4, 3] = pal[(bits >> (3 * i)) & 7]
    return out[:height, :width]


def zstd_supercompress(ktx2, level=6):
    """A scheme-0 single-level .ktx2 (this codec's UASTC output) rewritten the way stock `basisu -uastc -ktx2` writes it by default:
    supercompressionScheme 2, the level's data as ONE Zstandard frame made by the system's libzstd (ctypes; None when it is not installed),
    levelIndex byteLength = compressed size, uncompressedByteLength = original size (KTX 2.0 section 'supercompressionScheme')."""
    import ctypes as C, ctypes.util
    try:
        Z = C.CDLL("libzstd.so.1")
    except OSError:
        return None
    Z.ZSTD_compressBound.restype = C.c_size_t; Z.ZSTD_compressBound.argtypes = [C.c_size_t]
    Z.ZSTD_compress.restype = C.c_size_t; Z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    Z.ZSTD_isError.restype = C.c_uint; Z.ZSTD_isError.argtypes = [C.c_size_t]
    b = bytes(ktx2)
    assert int.from_bytes(b[44:48], "little") == 0 and int.from_bytes(b[40:44], "little") == 1            # scheme 0, one level
    lo, ll = int.from_bytes(b[80:88], "little"), int.from_bytes(b[88:96], "little")
    src = b[lo:lo + ll]
    cap = Z.ZSTD_compressBound(len(src)); buf = C.create_string_buffer(cap)
    n = Z.ZSTD_compress(buf, cap, src, len(src), level)
    assert not Z.ZSTD_isError(n)
    head = bytearray(b[:lo])
    head[44:48] = (2).to_bytes(4, "little"); head[88:96] = int(n).to_bytes(8, "little"); head[96:104] = len(src).to_bytes(8, "little")
    return bytes(head) + buf.raw[:n]


def psnr_rgb(a, b):
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    mse = float(np.mean(d * d))
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)
