"""oracle/oracle.py — ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (universal-volumetric_amd/) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


class OrcBuf(C.Structure):
    _fields_ = [("p", C.POINTER(C.c_uint8)), ("n", C.c_size_t), ("cap", C.c_size_t)]


class SymInfo(C.Structure):
    _fields_ = [("scheme", C.c_int), ("bl", C.c_int), ("prec_bits", C.c_int), ("alphabet", C.c_int),
                ("unique", C.c_int), ("left", C.c_int), ("final_state", C.c_uint32), ("base", C.c_uint32),
                ("payload", C.c_size_t)]


class DrcAtt(C.Structure):
    _fields_ = [("att_type", C.c_int), ("data_type", C.c_int), ("ncomp", C.c_int), ("unique_id", C.c_int),
                ("dec_type", C.c_int), ("att_data_id", C.c_int), ("seq_type", C.c_int),
                ("pred_method", C.c_int), ("transform", C.c_int), ("n", C.c_int), ("ncomp_port", C.c_int),
                ("vals", C.POINTER(C.c_int32)), ("corner_to_entry", C.POINTER(C.c_int32)),
                ("minv", C.c_float * 4), ("range", C.c_float), ("qbits", C.c_int),
                ("sec_begin", C.c_size_t), ("sec_end", C.c_size_t), ("sym_begin", C.c_size_t), ("sym_end", C.c_size_t),
                ("n_orient", C.c_int), ("n_flip_set", C.c_int), ("n_seam_corners", C.c_uint32)]


class DrcMesh(C.Structure):
    _fields_ = [("major", C.c_int), ("minor", C.c_int), ("nf", C.c_int), ("nev", C.c_int), ("nad", C.c_int),
                ("nsym", C.c_int), ("nsplit", C.c_int), ("nts", C.c_int), ("nverts_alloc", C.c_int),
                ("opp", C.POINTER(C.c_int32)), ("c2v", C.POINTER(C.c_int32)),
                ("ctx_n", C.c_int * 6), ("n_interior_start", C.c_int),
                ("conn_end", C.c_size_t), ("hdr_end", C.c_size_t), ("total", C.c_size_t),
                ("natt", C.c_int), ("att", DrcAtt * 8), ("leftover", C.c_size_t), ("method", C.c_int), ("traversal", C.c_int), ("npoints", C.c_int)]


class DrcEncParams(C.Structure):
    _fields_ = [("qp", C.c_int), ("qt", C.c_int), ("qn", C.c_int), ("method", C.c_int)]


class DrcEncInput(C.Structure):
    _fields_ = [("pos", C.POINTER(C.c_float)), ("n_pos", C.c_uint32),
                ("uv", C.POINTER(C.c_float)), ("n_uv", C.c_uint32),
                ("nrm", C.POINTER(C.c_float)), ("n_nrm", C.c_uint32),
                ("idx_pos", C.POINTER(C.c_uint32)), ("idx_uv", C.POINTER(C.c_uint32)), ("idx_nrm", C.POINTER(C.c_uint32)),
                ("nf", C.c_uint32)]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_void_p, C.c_size_t]
        L.drc_decode.restype = C.c_int
        L.drc_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(DrcMesh)]
        L.drc_mesh_free.argtypes = [C.POINTER(DrcMesh)]
        L.drc_dequant.argtypes = [C.POINTER(DrcMesh), C.c_int, C.POINTER(C.c_float)]
        L.orc_decode_symbols.restype = C.c_int
        L.orc_decode_symbols.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_uint32,
                                         C.POINTER(C.c_uint32), C.POINTER(SymInfo)]
        L.orc_encode_symbols.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(OrcBuf)]
        L.orc_rabs_encode.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(OrcBuf)]
        if hasattr(L, "drc_encode"):
            L.drc_encode.restype = C.c_int
            L.drc_encode.argtypes = [C.POINTER(DrcEncInput), C.POINTER(DrcEncParams), C.POINTER(OrcBuf)]
    return _LIB


def _take(buf):
    data = bytes(C.string_at(buf.p, buf.n)) if buf.n else b""
    C.CDLL(None).free(buf.p)
    return data


ATT_NAMES = {0: "position", 1: "normal", 2: "color", 3: "tex_coord", 4: "generic"}


class DecodedMesh:
    """Plain-numpy view of a decoded .drc (copies out of the C struct)."""

    def __init__(self, m):
        nf = m.nf
        self.nf, self.nev, self.nad, self.nsym, self.nsplit, self.nts = nf, m.nev, m.nad, m.nsym, m.nsplit, m.nts
        self.nverts_alloc = m.nverts_alloc
        self.ctx_n = list(m.ctx_n)
        self.n_interior_start = m.n_interior_start
        self.conn_end, self.hdr_end, self.leftover = m.conn_end, m.hdr_end, m.leftover
        self.method, self.traversal, self.npoints = m.method, m.traversal, m.npoints     # 1 edgebreaker / 0 sequential connectivity
        self.opp = np.ctypeslib.as_array(m.opp, (3 * nf,)).copy() if m.opp else None
        self.c2v = np.ctypeslib.as_array(m.c2v, (3 * nf,)).copy() if m.c2v else None
        self.atts = []
        for i in range(m.natt):
            a = m.att[i]
            d = {k: getattr(a, k) for k in ("att_type", "data_type", "ncomp", "unique_id", "dec_type", "att_data_id",
                                            "seq_type", "pred_method", "transform", "n", "ncomp_port", "range", "qbits",
                                            "sec_begin", "sec_end", "sym_begin", "sym_end", "n_orient", "n_flip_set",
                                            "n_seam_corners")}
            d["minv"] = list(a.minv)
            d["vals"] = np.ctypeslib.as_array(a.vals, (a.n * a.ncomp_port,)).copy().reshape(a.n, a.ncomp_port)
            d["corner_to_entry"] = np.ctypeslib.as_array(a.corner_to_entry, (3 * nf,)).copy()
            out = np.zeros(a.n * (3 if a.seq_type == 3 else a.ncomp), dtype=np.float32)
            lib().drc_dequant(C.byref(m), i, out.ctypes.data_as(C.POINTER(C.c_float)))
            d["float"] = out.reshape(a.n, -1)
            d["name"] = ATT_NAMES.get(a.att_type, str(a.att_type))
            self.atts.append(d)

    def att(self, name):
        for a in self.atts:
            if a["name"] == name:
                return a
        return None


def drc_decode(data: bytes) -> DecodedMesh:
    m = DrcMesh()
    rc = lib().drc_decode(data, len(data), C.byref(m))
    if rc != 0:
        raise ValueError(f"drc_decode failed rc={rc}")
    try:
        return DecodedMesh(m)
    finally:
        lib().drc_mesh_free(C.byref(m))


def crc32(arr) -> int:
    a = np.ascontiguousarray(arr)
    return lib().orc_crc32(a.ctypes.data, a.nbytes)


def decode_symbols(data: bytes, off: int, nvals: int):
    o = C.c_size_t(off)
    out = np.zeros(max(nvals, 1), dtype=np.uint32)
    info = SymInfo()
    rc = lib().orc_decode_symbols(data, len(data), C.byref(o), nvals, out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(info))
    if rc:
        raise ValueError(f"orc_decode_symbols rc={rc}")
    return out[:nvals], o.value, info


def encode_symbols(syms) -> bytes:
    s = np.ascontiguousarray(syms, dtype=np.uint32)
    buf = OrcBuf()
    lib().orc_encode_symbols(s.ctypes.data_as(C.POINTER(C.c_uint32)), len(s), C.byref(buf))
    return _take(buf)


def rabs_encode(bits) -> bytes:
    s = np.ascontiguousarray(bits, dtype=np.uint8)
    buf = OrcBuf()
    lib().orc_rabs_encode(s.ctypes.data_as(C.POINTER(C.c_uint8)), len(s), C.byref(buf))
    return _take(buf)


def drc_encode(pos, idx_pos, uv=None, idx_uv=None, nrm=None, idx_nrm=None, qp=11, qt=10, qn=8, method=0) -> bytes:
    """method: 0 valence edgebreaker (`draco_encoder -cl 7`), 1 edgebreaker with the standard traversal, 2 sequential connectivity, 3 sequential connectivity with compressed indices (connectivity_method 0; decoder test streams)."""
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    idx_pos = np.ascontiguousarray(idx_pos, dtype=np.uint32).reshape(-1)
    inp = DrcEncInput()
    keep = [pos, idx_pos]
    fp = C.POINTER(C.c_float)
    up = C.POINTER(C.c_uint32)
    inp.pos = pos.ctypes.data_as(fp); inp.n_pos = len(pos)
    inp.idx_pos = idx_pos.ctypes.data_as(up); inp.nf = len(idx_pos) // 3
    if uv is not None:
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        idx_uv = np.ascontiguousarray(idx_uv, dtype=np.uint32).reshape(-1)
        keep += [uv, idx_uv]
        inp.uv = uv.ctypes.data_as(fp); inp.n_uv = len(uv); inp.idx_uv = idx_uv.ctypes.data_as(up)
    if nrm is not None:
        nrm = np.ascontiguousarray(nrm, dtype=np.float32).reshape(-1, 3)
        idx_nrm = np.ascontiguousarray(idx_nrm, dtype=np.uint32).reshape(-1)
        keep += [nrm, idx_nrm]
        inp.nrm = nrm.ctypes.data_as(fp); inp.n_nrm = len(nrm); inp.idx_nrm = idx_nrm.ctypes.data_as(up)
    prm = DrcEncParams(qp, qt, qn, method)
    buf = OrcBuf()
    rc = lib().drc_encode(C.byref(inp), C.byref(prm), C.byref(buf))
    if rc:
        raise ValueError(f"drc_encode rc={rc}")
    return _take(buf)


# ------------------------------------------------------------------------------------------------
# KTX2 / BasisLZ ETC1S
# ------------------------------------------------------------------------------------------------
KTX2_MAX_LAYERS = 64


class Ktx2File(C.Structure):
    _fields_ = [("vk_format", C.c_uint32), ("type_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("depth", C.c_uint32), ("layers", C.c_uint32), ("faces", C.c_uint32), ("levels", C.c_uint32), ("supercomp", C.c_uint32),
                ("dfd_off", C.c_uint32), ("dfd_len", C.c_uint32), ("kvd_off", C.c_uint32), ("kvd_len", C.c_uint32),
                ("sgd_off", C.c_uint64), ("sgd_len", C.c_uint64), ("level_off", C.c_uint64), ("level_len", C.c_uint64), ("level_ulen", C.c_uint64),
                ("dfd_model", C.c_uint32), ("dfd_transfer", C.c_uint32), ("dfd_primaries", C.c_uint32),
                ("n_endpoints", C.c_uint32), ("n_selectors", C.c_uint32), ("endpoints_len", C.c_uint32), ("selectors_len", C.c_uint32),
                ("tables_len", C.c_uint32), ("extended_len", C.c_uint32), ("bx", C.c_uint32), ("by", C.c_uint32),
                ("endpoints", C.POINTER(C.c_uint8)), ("selectors", C.POINTER(C.c_uint32)), ("hist_size", C.c_uint32),
                ("ep_bits_used", C.c_uint32), ("sel_bits_used", C.c_uint32), ("tab_bits_used", C.c_uint32), ("n_slices", C.c_int),
                ("slice_flags", C.c_uint32 * KTX2_MAX_LAYERS), ("slice_off", C.c_uint32 * KTX2_MAX_LAYERS), ("slice_len", C.c_uint32 * KTX2_MAX_LAYERS),
                ("slice_bits_used", C.c_uint64 * KTX2_MAX_LAYERS), ("slice_skip", C.c_uint32 * KTX2_MAX_LAYERS),
                ("block_ei", C.POINTER(C.c_uint16)), ("block_si", C.POINTER(C.c_uint16)),
                ("writer", C.c_char * 64), ("anim_duration", C.c_uint32), ("anim_timescale", C.c_uint32), ("anim_loops", C.c_uint32), ("has_anim", C.c_int), ("has_alpha", C.c_int)]


class Ktx2EncParams(C.Structure):
    _fields_ = [("quality", C.c_int), ("y_flip", C.c_int)]


def _ktx2_setup():
    L = lib()
    if getattr(L, "_ktx2_ready", False):
        return L
    L.ktx2_decode.restype = C.c_int
    L.ktx2_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Ktx2File)]
    L.ktx2_free.argtypes = [C.POINTER(Ktx2File)]
    L.ktx2_layer_rgba.argtypes = [C.POINTER(Ktx2File), C.c_int, C.c_void_p]
    if hasattr(L, "ktx2_encode"):
        L.ktx2_encode.restype = C.c_int
        L.ktx2_encode.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(Ktx2EncParams), C.POINTER(OrcBuf)]
    L._ktx2_ready = True
    return L


class DecodedKtx2:
    def __init__(self, f, images=True):
        for k in ("vk_format", "type_size", "width", "height", "depth", "layers", "faces", "levels", "supercomp", "dfd_off", "dfd_len",
                  "kvd_off", "kvd_len", "sgd_off", "sgd_len", "level_off", "level_len", "level_ulen", "dfd_model", "dfd_transfer",
                  "dfd_primaries", "n_endpoints", "n_selectors", "endpoints_len", "selectors_len", "tables_len", "extended_len",
                  "bx", "by", "hist_size", "ep_bits_used", "sel_bits_used", "tab_bits_used", "n_slices", "has_anim",
                  "anim_duration", "anim_timescale", "anim_loops", "has_alpha"):
            setattr(self, k, getattr(f, k))
        self.writer = f.writer.decode(errors="replace")
        n = f.n_slices
        self.slice_flags = list(f.slice_flags[:n]); self.slice_off = list(f.slice_off[:n]); self.slice_len = list(f.slice_len[:n])
        self.slice_bits_used = list(f.slice_bits_used[:n]); self.slice_skip = list(f.slice_skip[:n])
        self.endpoints = np.ctypeslib.as_array(f.endpoints, (f.n_endpoints, 4)).copy()
        self.selectors = np.ctypeslib.as_array(f.selectors, (f.n_selectors,)).copy()
        nb = f.bx * f.by
        self.block_ei = np.ctypeslib.as_array(f.block_ei, (n, nb)).copy()
        self.block_si = np.ctypeslib.as_array(f.block_si, (n, nb)).copy()
        self.images = []
        if images:
            for l in range(max(1, f.layers)):               # one image per layer (its colour slice + its alpha slice, if any)
                img = np.zeros((f.height, f.width, 4), dtype=np.uint8)
                lib().ktx2_layer_rgba(C.byref(f), l, img.ctypes.data)
                self.images.append(img)


def ktx2_decode(data: bytes, images=True) -> DecodedKtx2:
    L = _ktx2_setup()
    f = Ktx2File()
    rc = L.ktx2_decode(data, len(data), C.byref(f))
    if rc:
        raise ValueError(f"ktx2_decode rc={rc}")
    try:
        return DecodedKtx2(f, images)
    finally:
        L.ktx2_free(C.byref(f))


def ktx2_goldens(data: bytes) -> dict:
    d = ktx2_decode(data)
    return dict(size=len(data), width=d.width, height=d.height, layers=d.layers, supercomp=d.supercomp, dfd_model=d.dfd_model,
                sgd_off=d.sgd_off, sgd_len=d.sgd_len, level_off=d.level_off, level_len=d.level_len,
                n_endpoints=d.n_endpoints, n_selectors=d.n_selectors, endpoints_len=d.endpoints_len, selectors_len=d.selectors_len,
                tables_len=d.tables_len, hist_size=d.hist_size, ep_bits=d.ep_bits_used, sel_bits=d.sel_bits_used, tab_bits=d.tab_bits_used,
                slice_flags=d.slice_flags, slice_off=d.slice_off, slice_len=d.slice_len, slice_bits=d.slice_bits_used, slice_skip=d.slice_skip,
                writer=d.writer, first_endpoints=d.endpoints[:3].tolist(),
                crc_ei="%08x" % crc32(d.block_ei), crc_si="%08x" % crc32(d.block_si),
                crc_img=["%08x" % crc32(im) for im in d.images])


def ktx2_encode(layers, quality=128, y_flip=1) -> bytes:
    L = _ktx2_setup()
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in layers]
    h, w = arrs[0].shape[:2]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    prm = Ktx2EncParams(quality, y_flip)
    buf = OrcBuf()
    rc = L.ktx2_encode(ptrs, len(arrs), w, h, C.byref(prm), C.byref(buf))
    if rc:
        raise ValueError(f"ktx2_encode rc={rc}")
    return _take(buf)


def psnr(a, b):
    a = np.asarray(a, dtype=np.float64)[..., :3]; b = np.asarray(b, dtype=np.float64)[..., :3]
    mse = ((a - b) ** 2).mean()
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


# ------------------------------------------------------------------------------------------------
# UASTC LDR 4x4 + ASTC 4x4 transcode (oracle/uastc.c; parity unpinned against basisu, see its header)
# ------------------------------------------------------------------------------------------------
class UastcLBlock(C.Structure):
    _fields_ = [("mode", C.c_int), ("ccs", C.c_int), ("ep", C.c_uint8 * 8), ("w", C.c_uint8 * 32), ("solid", C.c_uint8 * 4),
                ("bc1_hint0", C.c_int), ("bc1_hint1", C.c_int), ("etc1_flip", C.c_int), ("etc1_diff", C.c_int), ("etc1_inten0", C.c_int),
                ("etc1_inten1", C.c_int), ("etc1_bias", C.c_int), ("etc2_hints", C.c_int), ("etc1_sel", C.c_int), ("etc1_base", C.c_uint8 * 3)]


def _uastc_setup():
    L = lib()
    if getattr(L, "_uastc_ready", False):
        return L
    L.uastc_unpack.argtypes = [C.c_void_p, C.POINTER(UastcLBlock)]
    L.uastc_pack.argtypes = [C.POINTER(UastcLBlock), C.c_void_p]
    L.uastc_decode_block.argtypes = [C.c_void_p, C.c_void_p]
    L.uastc_to_astc.argtypes = [C.c_void_p, C.c_void_p]
    L.astc_decode_block.argtypes = [C.c_void_p, C.c_void_p]
    L.uastc_encode_block.argtypes = [C.c_void_p, C.c_void_p]
    L.uastc_ktx2_encode.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(OrcBuf)]
    L.uastc_ktx2_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.uastc_ktx2_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p]
    L._uastc_ready = True
    return L


def uastc_encode_blocks(px):
    """px: [n, 16, 4] uint8 texels (i = 4*y + x) -> [n, 16] uint8 UASTC blocks."""
    L = _uastc_setup()
    px = np.ascontiguousarray(px, dtype=np.uint8).reshape(-1, 64)
    out = np.zeros((len(px), 16), np.uint8)
    for i in range(len(px)):
        L.uastc_encode_block(px[i].ctypes.data, out[i].ctypes.data)
    return out


def uastc_decode_blocks(blocks):
    L = _uastc_setup()
    b = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
    out = np.zeros((len(b), 16, 4), np.uint8)
    for i in range(len(b)):
        rc = L.uastc_decode_block(b[i].ctypes.data, out[i].ctypes.data)
        if rc:
            raise ValueError(f"uastc_decode_block rc={rc} (block {i})")
    return out


def uastc_unpack(block):
    L = _uastc_setup()
    b = np.ascontiguousarray(block, dtype=np.uint8).reshape(16)
    lb = UastcLBlock()
    rc = L.uastc_unpack(b.ctypes.data, C.byref(lb))
    if rc:
        raise ValueError(f"uastc_unpack rc={rc}")
    return lb


def uastc_to_astc_blocks(blocks):
    L = _uastc_setup()
    b = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
    out = np.zeros((len(b), 16), np.uint8)
    for i in range(len(b)):
        rc = L.uastc_to_astc(b[i].ctypes.data, out[i].ctypes.data)
        if rc:
            raise ValueError(f"uastc_to_astc rc={rc} (block {i})")
    return out


def astc_decode_blocks(blocks):
    L = _uastc_setup()
    b = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
    out = np.zeros((len(b), 16, 4), np.uint8)
    for i in range(len(b)):
        rc = L.astc_decode_block(b[i].ctypes.data, out[i].ctypes.data)
        if rc:
            raise ValueError(f"astc_decode_block rc={rc} (block {i})")
    return out


def uastc_ktx2_encode(layers, y_flip=1) -> bytes:
    L = _uastc_setup()
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in layers]
    h, w = arrs[0].shape[:2]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    buf = OrcBuf()
    rc = L.uastc_ktx2_encode(ptrs, len(arrs), w, h, y_flip, C.byref(buf))
    if rc:
        raise ValueError(f"uastc_ktx2_encode rc={rc}")
    return _take(buf)


def uastc_ktx2_info(data: bytes):
    L = _uastc_setup()
    w, h, n, lo, a = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_int()
    rc = L.uastc_ktx2_info(data, len(data), C.byref(w), C.byref(h), C.byref(n), C.byref(lo), C.byref(a))
    if rc:
        raise ValueError(f"uastc_ktx2_info rc={rc}")
    return dict(width=w.value, height=h.value, layers=n.value, level_off=lo.value, has_alpha=bool(a.value))


def uastc_ktx2_decode(data: bytes, target="rgba"):
    """target 'rgba': [layers, H, W, 4] uint8 (stored row order); 'astc' / 'bc7': [layers, by, bx, 16] uint8 ASTC 4x4 / BC7 blocks."""
    L = _uastc_setup()
    i = uastc_ktx2_info(data)
    if target == "rgba":
        out = np.zeros((i["layers"], i["height"], i["width"], 4), np.uint8)
    else:
        out = np.zeros((i["layers"], (i["height"] + 3) // 4, (i["width"] + 3) // 4, 16), np.uint8)
    rc = L.uastc_ktx2_decode(data, len(data), {"rgba": 0, "astc": 1, "bc7": 2}[target], out.ctypes.data)
    if rc:
        raise ValueError(f"uastc_ktx2_decode rc={rc}")
    return out
