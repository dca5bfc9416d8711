"""Host-side mirror of scripts/Encoder.py (config rules, patterns, frame accounting, manifest, OBJ/PNG ingest)
against golden vectors recorded from the reference driver itself (tools/gen_golden_harness.py)."""
import ctypes as C
import json
import os
import subprocess
import numpy as np
import pytest
from conftest import ROOT, GOLDEN, REF_OUT

G = json.load(open(os.path.join(GOLDEN, "harness", "encoder_py_goldens.json")))


@pytest.fixture(scope="module")
def H():
    pkg = os.path.join(ROOT, "universal-volumetric_amd")
    subprocess.check_call(["make", "-s", "-C", pkg, "libuvolhost.so"])
    L = C.CDLL(os.path.join(pkg, "libuvolhost.so"))
    for n in ("uvolh_convert_pounds", "uvolh_format_index", "uvolh_check_config", "uvolh_check_total_frames", "uvolh_manifest", "uvolh_template"):
        getattr(L, n).restype = C.c_char_p
    L.uvolh_check_total_frames.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_double, C.c_double]
    L.uvolh_manifest.argtypes = [C.c_char_p, C.c_long, C.c_long, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    L.uvolh_format_index.argtypes = [C.c_char_p, C.c_uint]
    return L


def test_convert_pounds(H):
    for s, want in G["convert_pounds_to_c_style"].items():
        assert H.uvolh_convert_pounds(s.encode()).decode() == want


def test_match_pattern_truth_table(H):
    for p, f, want in G["match_pattern"]:
        assert bool(H.uvolh_match_pattern(p.encode(), f.encode())) == want, (p, f)
    # the lenient form additionally accepts bracketed patterns against bare file names
    assert H.uvolh_match_pattern_lenient(b"frame_[#######].obj", b"frame_0000001.obj") == 1
    assert H.uvolh_format_index(b"geometry/[#####].drc", 42).decode() == "geometry/00042.drc"


def test_check_all_fields_matrix(H):
    for name, case in G["check_all_fields"].items():
        got = H.uvolh_check_config(json.dumps(case["config"]).encode()).decode()
        want = case["message"]
        assert (("❌ " + got) if got else "") == want, name


def test_config_with_comments_and_template(H):
    t = H.uvolh_template().decode()
    assert '"KTX2_BATCH_SIZE": 7' in t and "// quantization bits for the position attribute, default=11." in t
    # the template (comments included) parses; it is rejected for the same reason the reference rejects it (no geometry path)
    assert H.uvolh_check_config(t.encode()).decode() == "Path to Geometry data is not specified"


@pytest.mark.skipif(not os.path.isdir(REF_OUT), reason="/root/reference only exists in the build container")
def test_check_total_frames_on_reference_fixture(H):
    g = G["check_total_frames_fixture"]
    r = json.loads(H.uvolh_check_total_frames(os.path.join(REF_OUT, "geometry_draco", "#####.drc").encode(),
                                              os.path.join(REF_OUT, "texture_ktx2-fps30-1k_baseColor_default", "#####.ktx2").encode(), 5, 30.0, 30.0))
    assert (r["geometry_frames"], r["segments"], r["compatible"]) == (g["geometry_frames"], g["segments"], True)
    assert r["geometry"] == g["durations"]["geometry"] and r["texture"] == g["durations"]["texture"]


def test_check_total_frames_short_last_segment(H, tmp_path):
    (tmp_path / "g").mkdir(); (tmp_path / "t").mkdir()
    for i in range(7):
        (tmp_path / "g" / ("%05d.drc" % i)).write_bytes(b"x")
    hdr = bytearray(open(os.path.join(GOLDEN, "00000.ktx2"), "rb").read(80))
    (tmp_path / "t" / "00000.ktx2").write_bytes(bytes(hdr))
    hdr[32:36] = (2).to_bytes(4, "little")
    (tmp_path / "t" / "00001.ktx2").write_bytes(bytes(hdr))
    r = json.loads(H.uvolh_check_total_frames(str(tmp_path / "g" / "#####.drc").encode(), str(tmp_path / "t" / "#####.ktx2").encode(), 5, 30.0, 30.0))
    assert (r["geometry_frames"], r["texture_frames"], r["segments"], r["compatible"]) == (7, 7, 2, True)


CFG = {"name": "n", "GEOMETRY_FRAME_RATE": 30, "TEXTURE_FRAME_RATE": 30, "OutputDirectory": "out", "KTX2_BATCH_SIZE": 5,
       "OBJFilesPath": "OBJ/f_#####.obj", "ImagesPath": "PNG/t_#####.png", "KTX2_FIRST_FILE": 0, "KTX2_FILE_COUNT": 250}


def test_manifest_encoder_py_shape(H):
    m = json.loads(H.uvolh_manifest(json.dumps(CFG).encode(), 250, 50, 1024, 1024, 5, 1, b"DRC/#####.drc", b"KTX2/#####.ktx2"))
    assert m == G["manifest_encoder_py"]


def test_manifest_is_playable_by_stock_player_url_templating(H):
    """Re-implements getGeometryURL / getTextureURL of src/V2/player.ts:141-174 and playTrack's target selection (:207-222)."""
    m = json.loads(H.uvolh_manifest(json.dumps(CFG).encode(), 250, 50, 2048, 2048, 5, 0, b"", b""))
    FORMATS_TO_EXT = {"mp3": ".mp3", "draco": ".drc", "ktx2": ".ktx2", "etc2": ".etc2"}
    assert m["version"] == "v2"
    gt = list(m["geometry"]["targets"].keys())[0]             # Object.keys(...)[0] needs an OBJECT
    tt = list(m["texture"]["targets"].keys())[0]
    assert tt in ("ktx2", "mp4")                               # isTextureFormatSupported(renderer, key)

    def url(path, inputs, n):
        pad = path.count("#")
        inputs["[" + "#" * pad + "]"] = str(n).rjust(pad, "0")
        for k, v in inputs.items():
            path = path.replace(k, v, 1)
        return path
    assert url(m["geometry"]["path"], {"[target]": gt, "[ext]": FORMATS_TO_EXT[m["geometry"]["targets"][gt]["format"]]}, 7) == "geometry_draco/00007.drc"
    assert url(m["texture"]["path"], {"[target]": tt, "[type]": "baseColor", "[tag]": "default", "[ext]": FORMATS_TO_EXT[m["texture"]["targets"][tt]["format"]]}, 3) == "texture_ktx2_baseColor_default/00003.ktx2"
    t = m["texture"]["targets"][tt]
    assert (t["sequenceSize"], t["sequenceCount"], t["frameRate"], t["resolution"]) == (5, 50, 30, [2048, 2048]) and t["sequenceSize"] >= 2
    assert m["geometry"]["targets"][gt] == {"format": "draco", "frameRate": 30, "frameCount": 250}


def test_manifest_urls_by_node_player_restatement(H, tmp_path):
    """Same check through tests/player_urls.js (node 12): the manifest parses as v2 and yields the file names uvolenc writes."""
    import shutil, subprocess
    if not shutil.which("node"):
        pytest.skip("node not available")
    mp = tmp_path / "uvol.json"
    mp.write_bytes(H.uvolh_manifest(json.dumps(CFG).encode(), 12, 3, 64, 64, 5, 0, b"", b""))
    urls = json.loads(subprocess.check_output(["node", os.path.join(ROOT, "tests", "player_urls.js"), str(mp)], text=True))
    assert urls["geometry"] == ["geometry_draco/%05d.drc" % k for k in range(12)]
    assert urls["texture"] == ["texture_ktx2_baseColor_default/%05d.ktx2" % k for k in range(3)]
    assert (urls["geometryTarget"], urls["textureTarget"], urls["batchSize"]) == ("draco", "ktx2", 5)


def test_obj_and_png_ingest(H, tmp_path):
    import synth
    from PIL import Image
    m = synth.torus_mesh(12, 6)
    p = tmp_path / "a.obj"
    with open(p, "w") as f:
        f.write("# comment\nmtllib x.mtl\n")
        for v in m["pos"]: f.write("v %r %r %r\n" % tuple(float(x) for x in v))
        for v in m["uv"]: f.write("vt %r %r\n" % tuple(float(x) for x in v))
        for v in m["nrm"]: f.write("vn %r %r %r\n" % tuple(float(x) for x in v))
        ip, iu, inn = (m[k].reshape(-1, 3) + 1 for k in ("idx_pos", "idx_uv", "idx_nrm"))
        for a, b, c in zip(ip, iu, inn):
            f.write("f " + " ".join("%d/%d/%d" % (a[k], b[k], c[k]) for k in range(3)) + "\n")
        f.write("f 1/1/1 2/2/2 3/3/3 4/4/4\n")            # a quad: fanned into two triangles
    out = (C.c_uint * 6)()
    assert H.uvolh_read_obj_counts(str(p).encode(), out) == 0
    nf = len(m["idx_pos"]) // 3 + 2
    assert list(out) == [len(m["pos"]), len(m["uv"]), len(m["nrm"]), nf, nf, nf]
    rng = np.random.default_rng(0)
    for mode, arr in (("RGBA", rng.integers(0, 256, (13, 17, 4), dtype=np.uint8)), ("RGB", rng.integers(0, 256, (9, 5, 3), dtype=np.uint8)), ("L", rng.integers(0, 256, (6, 7), dtype=np.uint8))):
        q = tmp_path / (mode + ".png"); Image.fromarray(arr, mode).save(q)
        wh = (C.c_uint * 2)(); buf = np.zeros((arr.shape[0], arr.shape[1], 4), np.uint8)
        assert H.uvolh_read_png(str(q).encode(), wh, buf.ctypes.data_as(C.POINTER(C.c_ubyte)), buf.nbytes) == 0
        want = np.array(Image.fromarray(arr, mode).convert("RGBA"))
        assert (wh[0], wh[1]) == (arr.shape[1], arr.shape[0]) and np.array_equal(buf, want)


def test_obj_number_parser_equals_strtof(H, tmp_path):
    """read_obj parses decimals itself (strtof is ~10x slower); the result must be strtof's, bit for bit: plain decimals of every
    length, exponents, signs, leading / trailing dots, values next to float rounding boundaries, subnormals, overflow, long digit strings."""
    import ctypes.util
    libc = C.CDLL(ctypes.util.find_library("c")); libc.strtof.restype = C.c_float; libc.strtof.argtypes = [C.c_char_p, C.c_void_p]
    rng = np.random.default_rng(5)
    toks = []
    for k in range(3000):
        x = float(rng.standard_normal()) * 10.0 ** int(rng.integers(-12, 12))
        toks += ["%r" % x, "%.6f" % x, "%.3g" % x, "%.9g" % x, "%.17g" % x, "%e" % x]
    f32 = rng.integers(0, 2 ** 31 - 2 ** 23, size=2000, dtype=np.int64).astype(np.uint32).view(np.float32)      # finite floats incl. subnormals
    for v in f32:
        d = float(v); nx = float(np.nextafter(v, np.float32(np.inf)))
        toks += ["%.9g" % d, "%.17g" % ((d + nx) / 2), "%.20g" % ((d + nx) / 2), "%.12g" % ((d + nx) / 2)]   # the midpoints: rounding boundaries
    toks += ["0", "-0", "-0.0", "+1.5", ".5", "5.", "-.25", "1e-40", "1e-46", "3.4028235e38", "3.5e38", "1e39", "-1e39", "123456789012345678901234567890",
             "0.000000000000000000000000000000000000001", "1.00000005960464477539062500000000000001", "16777217", "16777217.0000000001", "9007199254740993", "1e22", "1e23", "8.5e-23"]
    toks = toks[: 3 * (len(toks) // 3)]
    p = tmp_path / "n.obj"
    with open(p, "w") as f:
        for i in range(0, len(toks), 3):
            f.write("v %s %s %s\n" % tuple(toks[i:i + 3]))
        f.write("f 1 2 3\n")
    H.uvolh_read_obj_positions.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_size_t]
    got = np.zeros(len(toks), np.float32)
    assert H.uvolh_read_obj_positions(str(p).encode(), got.ctypes.data_as(C.POINTER(C.c_float)), got.size) == len(toks) // 3
    want = np.array([libc.strtof(t.encode(), None) for t in toks], np.float32)
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    assert len(bad) == 0, [(toks[i], got[i], want[i]) for i in bad[:5]]


def test_png_filters_modes_and_both_inflaters(tmp_path):
    """read_png un-filters rows in place with one loop per filter type and inflates through libdeflate when the shared library is there,
    else zlib: every PNG filter type (rows written by hand), 8- and 16-bit, grey / grey+alpha / RGB / RGBA / palette with tRNS, and both
    inflaters (UVOL_NO_LIBDEFLATE=1 in a child process: the choice is made once per process) must give PIL's pixels."""
    import struct, subprocess, sys, zlib
    from PIL import Image
    rng = np.random.default_rng(11)
    h, w = 37, 53
    base = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2)[..., None] + rng.integers(0, 9, (h, w, 4))).astype(np.uint8)      # smooth + noise: all filters pay

    def png_with_filters(arr):                                           # arr [h, w, c] uint8; row y uses filter y % 5
        hh, ww, c = arr.shape; raw = bytearray(); prev = np.zeros(ww * c, np.int32)
        for y in range(hh):
            cur = arr[y].reshape(-1).astype(np.int32); ft = y % 5
            a = np.concatenate([np.zeros(c, np.int32), cur[:-c]]); b = prev; cc = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
            if ft == 0: f = cur
            elif ft == 1: f = cur - a
            elif ft == 2: f = cur - b
            elif ft == 3: f = cur - (a + b) // 2
            else:
                p = a + b - cc; pa, pb, pc = abs(p - a), abs(p - b), abs(p - cc)
                f = cur - np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, cc))
            raw.append(ft); raw += bytes((f & 255).astype(np.uint8)); prev = cur
        def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
        ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
        z = zlib.compress(bytes(raw), 6)
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", ww, hh, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    files = []
    for c in (1, 2, 3, 4):
        p = tmp_path / ("filters%d.png" % c); p.write_bytes(png_with_filters(base[..., :c].copy())); files.append(p)
    for name, im in (("rgb16", Image.fromarray(rng.integers(0, 65536, (9, 11)).astype(np.uint16))),
                     ("pal", Image.fromarray(rng.integers(0, 256, (12, 13, 3), dtype=np.uint8), "RGB").quantize(17)),
                     ("la", Image.fromarray(rng.integers(0, 256, (8, 7, 2), dtype=np.uint8), "LA"))):
        p = tmp_path / (name + ".png")
        if name == "pal": im.save(p, transparency=3)
        else: im.save(p)
        files.append(p)
    prog = (
        "import sys, ctypes as C, numpy as np\n"
        "from PIL import Image\n"
        "H = C.CDLL(sys.argv[1])\n"
        "for f in sys.argv[2:]:\n"
        "    want = np.array(Image.open(f).convert('RGBA')) if Image.open(f).mode != 'I;16' else None\n"
        "    im = Image.open(f); wh = (C.c_uint * 2)(); buf = np.zeros((im.size[1], im.size[0], 4), np.uint8)\n"
        "    assert H.uvolh_read_png(f.encode(), wh, buf.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_size_t(buf.nbytes)) == 0, f\n"
        "    if want is None:\n"
        "        g = (np.array(im) >> 8).astype(np.uint8); want = np.stack([g, g, g, np.full_like(g, 255)], -1)\n"
        "    assert (wh[0], wh[1]) == im.size and np.array_equal(buf, want), f\n"
        "print('ok')\n")
    lib = os.path.join(ROOT, "universal-volumetric_amd", "libuvolhost.so")
    for env_extra in ({}, {"UVOL_NO_LIBDEFLATE": "1"}):
        out = subprocess.run([sys.executable, "-c", prog, lib] + [str(f) for f in files], env=dict(os.environ, **env_extra), capture_output=True, text=True)
        assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-800:]
