"""UASTC LDR 4x4 oracle (oracle/uastc.c).  No reference fixture holds a UASTC file and basisu is not in the image, so parity
with basisu is UNPINNED; what these tests pin is the internal consistency of the restated formats: two independently written
decoders (UASTC -> RGBA and ASTC -> RGBA through a generic integer-sequence decoder) agree on every block, the mode prefix codes
form a complete prefix code, every emitted mode fills exactly 128 bits, and the KTX2 container carries the fields the stock
player's loader checks (reference src/lib/KTX2Loader.js:297-343: vkFormat 0, DFD colour model 166 = UASTC)."""
import os
import shutil
import subprocess
import numpy as np
import pytest


def _blocks(img):
    h, w = img.shape[:2]
    return img.reshape(h // 4, 4, w // 4, 4, 4).transpose(0, 2, 1, 3, 4).reshape(-1, 16, 4)


def _corpus():
    import synth
    rng = np.random.default_rng(0)
    tex = _blocks(synth.texture_sequence(1, size=128, seed=3)[0])
    noise = rng.integers(0, 256, size=(150, 16, 4)).astype(np.uint8); noise[..., 3] = 255
    alpha = tex[:200].copy(); alpha[..., 3] = rng.integers(0, 256, size=(200, 16)).astype(np.uint8)
    solid = np.repeat(rng.integers(0, 256, size=(12, 1, 4)), 16, axis=1).astype(np.uint8)
    grad = np.zeros((64, 16, 4), np.uint8); grad[..., 3] = 255
    for i in range(64):
        for t in range(16):
            grad[i, t, :3] = [(i * 4 + (t % 4) * 3) % 256, (t // 4) * 20 + i, 255 - i * 3]
    two = np.zeros((32, 16, 4), np.uint8); two[..., 3] = 255; two[:, :8, 0] = 250; two[:, 8:, 2] = 250
    return dict(texture=tex, noise=noise, alpha=alpha, solid=solid, gradient=grad, two_colour=two)


def test_mode_prefix_code_is_complete_and_modes_fill_128_bits(oracle):
    huff = [(0x1, 4), (0x35, 6), (0x1D, 5), (0x3, 5), (0x13, 5), (0xB, 5), (0x1B, 5), (0x7, 5), (0x17, 5), (0xF, 5),
            (0x2, 3), (0x0, 2), (0x6, 3), (0x1F, 5), (0xD, 5), (0x5, 7), (0x15, 6), (0x25, 6), (0x9, 4), (0x45, 7)]
    assert sum(2.0 ** -l for _, l in huff) == 1.0                       # Kraft equality: a complete code
    for i, (a, la) in enumerate(huff):                                   # prefix-free, codes read LSB first
        for j, (b, lb) in enumerate(huff):
            if i != j and la <= lb:
                assert (b & ((1 << la) - 1)) != a
    wbits = [4, 2, 3, 2, 2, 3, 2, 2, 0, 2, 4, 2, 3, 1, 2, 4, 2, 2, 5]
    rng_ = [19, 20, 8, 7, 12, 20, 18, 12, 0, 8, 13, 13, 19, 20, 20, 20, 20, 20, 11]
    comps = [3, 3, 3, 3, 3, 3, 3, 3, 0, 4, 4, 4, 4, 4, 4, 2, 2, 2, 3]
    planes = [1, 1, 1, 1, 1, 1, 2, 1, 0, 1, 1, 2, 1, 2, 1, 1, 1, 2, 1]
    hints = [15, 15, 15, 15, 15, 15, 15, 15, 0, 23, 17, 17, 17, 23, 23, 23, 23, 23, 15]
    btq = {11: (5, 0, 0), 13: (4, 1, 0), 18: (5, 0, 1), 19: (6, 1, 0)}
    for m in (0, 6, 10, 11, 12, 18):                                     # the modes this codec emits
        b, t, q = btq[rng_[m]]; nv = 2 * comps[m]
        ep = nv * b + (t and (8 * nv + 4) // 5) + (q and (7 * nv + 2) // 3)
        total = huff[m][1] + hints[m] + (2 if planes[m] == 2 else 0) + ep + 16 * planes[m] * wbits[m] - planes[m]
        assert total == 128, (m, total)


def test_two_decoders_agree_and_quality(oracle):
    C = _corpus()
    for name, px in C.items():
        U = oracle.uastc_encode_blocks(px)
        D = oracle.uastc_decode_blocks(U)
        A = oracle.uastc_to_astc_blocks(U)
        assert np.array_equal(D, oracle.astc_decode_blocks(A)), name      # UASTC decode == independent ASTC decode of the transcode
        modes = {oracle.uastc_unpack(u).mode for u in U}
        assert modes <= {0, 6, 8, 10, 11, 12, 18}, (name, modes)
        mse = ((D[..., :4].astype(np.float64) - px.astype(np.float64)) ** 2).mean()
        psnr = 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))
        floor = dict(texture=40.0, gradient=44.0, solid=98.0, two_colour=27.0, alpha=33.0, noise=14.0)[name]
        assert psnr >= floor, (name, psnr)
    # the anchor rule: the first weight of every plane has its top bit clear
    for u in oracle.uastc_encode_blocks(C["texture"][:200]):
        lb = oracle.uastc_unpack(u)
        if lb.mode != 8:
            wb = {0: 4, 6: 2, 10: 4, 11: 2, 12: 3, 18: 5}[lb.mode]
            for p in range(2 if lb.mode in (6, 11) else 1):
                assert lb.w[p] < (1 << (wb - 1))


def test_astc_void_extent_and_block_modes(oracle):
    px = np.full((1, 16, 4), 0, np.uint8); px[0, :, :] = [12, 200, 33, 255]
    a = oracle.uastc_to_astc_blocks(oracle.uastc_encode_blocks(px))[0]
    assert bytes(a[:8]) == bytes.fromhex("fcfdffffffffffff")              # LDR void-extent header of the ASTC specification
    assert list(a[8:]) == [12, 12, 200, 200, 33, 33, 255, 255]
    # every transcoded block carries the block mode of its UASTC mode: 4x4 grid, single partition, CEM 8 / 12
    C = _corpus()
    U = oracle.uastc_encode_blocks(np.concatenate([C["texture"][:300], C["alpha"][:100]]))
    bm = {0: 0x242, 6: 0x442, 10: 0x242, 11: 0x442, 12: 0x53, 18: 0x253}
    for u, a in zip(U, oracle.uastc_to_astc_blocks(U)):
        m = oracle.uastc_unpack(u).mode
        if m == 8:
            continue
        lo = int(a[0]) | (int(a[1]) << 8) | (int(a[2]) << 16)
        assert lo & 0x7FF == bm[m] and (lo >> 11) & 3 == 0 and (lo >> 13) & 15 == (12 if m in (10, 11, 12) else 8)


def test_ktx2_container_and_roundtrip(oracle, tmp_path):
    import synth
    tex = synth.texture_sequence(3, size=64, seed=5)
    k = oracle.uastc_ktx2_encode(tex)
    info = oracle.uastc_ktx2_info(k)
    assert (info["width"], info["height"], info["layers"], info["has_alpha"]) == (64, 64, 3, False) and info["level_off"] % 16 == 0
    assert len(k) == info["level_off"] + 3 * 16 * 16 * 16
    hdr = np.frombuffer(k[12:48], np.uint32)
    assert list(hdr) == [0, 1, 64, 64, 0, 3, 1, 1, 0]                     # vkFormat 0, typeSize 1, ..., no supercompression
    dfd = int(np.frombuffer(k[48:52], np.uint32)[0])
    assert k[dfd + 12] == 166 and k[dfd + 14] == 2 and k[dfd + 20] == 16 and k[dfd + 30] == 127      # model UASTC, sRGB, 16-byte blocks, 128-bit sample
    dec = oracle.uastc_ktx2_decode(k)
    src = np.stack([np.asarray(a)[::-1] for a in tex])                    # -y_flip: stored bottom-up
    assert oracle.psnr(src, dec) > 36.0                                   # 64x64 multi-octave noise: the hardest content per block (43 dB at 256x256)
    a = tex[1].copy(); a[..., 3] = 100
    assert oracle.uastc_ktx2_info(oracle.uastc_ktx2_encode([tex[0], a]))["has_alpha"]
    # container cross-read with the reference's own KTX-Parse (src/lib/ktx-parse.module.js) when node is available
    node = shutil.which("node")
    ref = "/root/reference/src/lib/ktx-parse.module.js"
    if node and os.path.exists(ref):
        (tmp_path / "t.ktx2").write_bytes(k)
        shutil.copy(ref, tmp_path / "ktxparse.mjs")
        js = ("import { read } from './ktxparse.mjs'; import fs from 'fs';"
              "const c = read(new Uint8Array(fs.readFileSync('t.ktx2')));"
              "const d = c.dataFormatDescriptor[0];"
              "console.log(JSON.stringify({vk: c.vkFormat, w: c.pixelWidth, h: c.pixelHeight, layers: c.layerCount, sc: c.supercompressionScheme, model: d.colorModel, transfer: d.transferFunction, levels: c.levels.length, l0: c.levels[0].levelData.byteLength}))")
        (tmp_path / "r.mjs").write_text(js)
        r = subprocess.run([node, "--experimental-modules", "r.mjs"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr[-800:]
        import json
        o = json.loads(r.stdout.strip().splitlines()[-1])
        assert o == dict(vk=0, w=64, h=64, layers=3, sc=0, model=166, transfer=2, levels=1, l0=3 * 16 * 16 * 16)
