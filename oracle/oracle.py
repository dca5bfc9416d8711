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
                ("natt", C.c_int), ("att", DrcAtt * 8), ("leftover", C.c_size_t)]


class DrcEncParams(C.Structure):
    _fields_ = [("qp", C.c_int), ("qt", C.c_int), ("qn", C.c_int)]


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
        self.opp = np.ctypeslib.as_array(m.opp, (3 * nf,)).copy()
        self.c2v = np.ctypeslib.as_array(m.c2v, (3 * nf,)).copy()
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


def drc_encode(pos, idx_pos, uv=None, idx_uv=None, nrm=None, idx_nrm=None, qp=11, qt=10, qn=8) -> bytes:
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
    prm = DrcEncParams(qp, qt, qn)
    buf = OrcBuf()
    rc = lib().drc_encode(C.byref(inp), C.byref(prm), C.byref(buf))
    if rc:
        raise ValueError(f"drc_encode rc={rc}")
    return _take(buf)
