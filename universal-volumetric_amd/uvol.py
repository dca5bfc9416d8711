"""uvol.py — thin ctypes mirror of include/uvol_codec.h (libuvolcodec.so, HIP/gfx950).

Host-side counterpart of the two process boundaries of the reference driver
(scripts/Encoder.py:260-262 `draco_encoder`, :290-292 `basisu`).  There is no CPU fallback: if the
HIP library or a GPU is missing, `Codec()` raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libuvolcodec.so")

UVOL_OK, UVOL_E_INVALID, UVOL_E_NODEVICE, UVOL_E_HIP, UVOL_E_NOSPACE, UVOL_E_ENCODE, UVOL_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6


class DecodedMesh(C.Structure):
    _fields_ = [("cap_faces", C.c_uint32), ("cap_values", C.c_size_t), ("pos", C.c_void_p), ("uv", C.c_void_p), ("nrm", C.c_void_p),
                ("idx_pos", C.c_void_p), ("idx_uv", C.c_void_p), ("idx_nrm", C.c_void_p),
                ("n_faces", C.c_uint32), ("n_pos", C.c_uint32), ("n_uv", C.c_uint32), ("n_nrm", C.c_uint32)]


class Params(C.Structure):
    """project-config.json numeric fields used on the hot path (scripts/Encoder.py:171-179)."""
    _fields_ = [("Q_POSITION_ATTR", C.c_int32), ("Q_TEXTURE_ATTR", C.c_int32), ("Q_NORMAL_ATTR", C.c_int32),
                ("Q_GENERIC_ATTR", C.c_int32), ("DRACO_COMPRESSION_LEVEL", C.c_int32), ("KTX2_BATCH_SIZE", C.c_int32),
                ("etc1s_quality", C.c_int32), ("y_flip", C.c_int32), ("max_batch", C.c_int32), ("cu_mod", C.c_int32), ("cu_residues", C.c_int32),
                ("traverse_vbits_l2", C.c_int32), ("stream_priority", C.c_int32), ("uastc", C.c_int32), ("reserved", C.c_int32 * 2)]


class Mesh(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("n_pos", C.c_uint32), ("uv", C.c_void_p), ("n_uv", C.c_uint32),
                ("nrm", C.c_void_p), ("n_nrm", C.c_uint32), ("idx_pos", C.c_void_p), ("idx_uv", C.c_void_p),
                ("idx_nrm", C.c_void_p), ("n_faces", C.c_uint32)]


EXPORTS = ["uvol_params_default", "uvol_abi_version", "uvol_device_count", "uvol_ctx_create", "uvol_ctx_destroy",
           "uvol_last_error", "uvol_sync", "uvol_trim", "uvol_mesh_bound", "uvol_mesh_workspace", "uvol_encode_mesh", "uvol_encode_mesh_batch",
           "uvol_encode_mesh_batch_dev", "uvol_encode_mesh_batch_dev_out", "uvol_decode_mesh_batch_dev", "uvol_parse_obj_batch_dev", "uvol_unfilter_png_batch_dev", "uvol_encode_mesh_batch_async", "uvol_encode_mesh_batch_dev_async", "uvol_encode_texture_segments_async", "uvol_encode_texture_segments_dev_async", "uvol_texture_bound", "uvol_encode_texture_segment",
           "uvol_encode_texture_segment_dev", "uvol_encode_texture_segments", "uvol_encode_texture_segments_dev",
           "uvol_ktx2_info", "uvol_decode_texture_segments", "uvol_decode_texture_segments_dev", "uvol_transcode_texture_segments_etc1", "uvol_transcode_texture_segments_bc7", "uvol_transcode_texture_segments_etc2_rgba", "uvol_transcode_texture_segments_astc", "uvol_drc_info", "uvol_decode_mesh_batch", "uvol_profile_enable", "uvol_profile_reset", "uvol_profile_count",
           "uvol_profile_get", "uvol_encode_texture_segments_st", "uvol_transcode_texture_segments_st",
           "uvol_host_alloc", "uvol_host_free", "uvol_inflate_png_batch_dev", "uvol_png_status"]


def load(path=None):
    path = path or os.environ.get("UVOL_LIB") or DEFAULT_LIB      # UVOL_LIB: diagnostic builds of the same HIP library (tools/)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950); no CPU fallback exists")
    L = C.CDLL(path)
    L.uvol_params_default.argtypes = [C.POINTER(Params)]
    L.uvol_ctx_create.argtypes = [C.c_int, C.POINTER(Params), C.POINTER(C.c_void_p)]
    L.uvol_ctx_destroy.argtypes = [C.c_void_p]
    L.uvol_last_error.argtypes = [C.c_void_p]; L.uvol_last_error.restype = C.c_char_p
    L.uvol_sync.argtypes = [C.c_void_p]
    L.uvol_trim.argtypes = [C.c_void_p]
    L.uvol_mesh_bound.argtypes = [C.POINTER(Mesh)]; L.uvol_mesh_bound.restype = C.c_size_t
    L.uvol_mesh_workspace.argtypes = [C.c_void_p, C.POINTER(Mesh)]; L.uvol_mesh_workspace.restype = C.c_size_t
    L.uvol_encode_mesh.argtypes = [C.c_void_p, C.POINTER(Mesh), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    for nm in ("uvol_encode_mesh_batch", "uvol_encode_mesh_batch_dev", "uvol_encode_mesh_batch_async", "uvol_encode_mesh_batch_dev_async"):
        getattr(L, nm).argtypes = [C.c_void_p, C.POINTER(Mesh), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    L.uvol_texture_bound.argtypes = [C.c_uint32, C.c_uint32, C.c_int]; L.uvol_texture_bound.restype = C.c_size_t
    for nm in ("uvol_encode_texture_segment", "uvol_encode_texture_segment_dev"):
        getattr(L, nm).argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p,
                                   C.c_size_t, C.POINTER(C.c_size_t)]
    for nm in ("uvol_encode_texture_segments", "uvol_encode_texture_segments_dev", "uvol_encode_texture_segments_async", "uvol_encode_texture_segments_dev_async"):
        getattr(L, nm).argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.uvol_ktx2_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    for nm in ("uvol_decode_texture_segments", "uvol_decode_texture_segments_dev"):
        getattr(L, nm).argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_void_p), C.c_size_t]
    L.uvol_transcode_texture_segments_etc1.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_int]
    L.uvol_transcode_texture_segments_bc7.argtypes = L.uvol_transcode_texture_segments_etc1.argtypes
    L.uvol_transcode_texture_segments_astc.argtypes = L.uvol_transcode_texture_segments_etc1.argtypes
    L.uvol_transcode_texture_segments_etc2_rgba.argtypes = L.uvol_transcode_texture_segments_etc1.argtypes
    L.uvol_encode_texture_segments_st.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p),
                                                  C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    L.uvol_transcode_texture_segments_st.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.uvol_host_alloc.argtypes = [C.c_size_t]; L.uvol_host_alloc.restype = C.c_void_p
    L.uvol_host_free.argtypes = [C.c_void_p]
    L.uvol_drc_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.uvol_decode_mesh_batch.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(DecodedMesh), C.POINTER(C.c_int)]
    L.uvol_decode_mesh_batch_dev.argtypes = L.uvol_decode_mesh_batch.argtypes
    L.uvol_unfilter_png_batch_dev.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.uvol_inflate_png_batch_dev.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.uvol_png_status.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.uvol_parse_obj_batch_dev.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(Mesh), C.POINTER(C.c_int)]
    L.uvol_encode_mesh_batch_dev_out.argtypes = [C.c_void_p, C.POINTER(Mesh), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    L.uvol_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.uvol_profile_reset.argtypes = [C.c_void_p]
    L.uvol_profile_count.argtypes = [C.c_void_p]
    L.uvol_profile_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    return L


class UvolError(RuntimeError):
    pass


class PinnedArena:
    """Page-locked host memory from uvol_host_alloc, handed out as numpy arrays (the arrays must not outlive the arena).  Inputs that ALL lie
    in such memory are uploaded without the library's staging copy."""

    def __init__(self, nbytes, lib_path=None):
        self.L = load(lib_path); self.n = int(nbytes); self.off = 0
        self.p = self.L.uvol_host_alloc(self.n)
        if not self.p:
            raise UvolError(f"uvol_host_alloc({self.n}) failed")
        self.buf = (C.c_uint8 * self.n).from_address(self.p)

    def put(self, a):
        a = np.ascontiguousarray(a); nb = a.nbytes; o = (self.off + 255) & ~255
        if o + nb > self.n:
            raise UvolError("pinned arena full")
        v = np.frombuffer(self.buf, dtype=a.dtype, count=a.size, offset=o).reshape(a.shape)
        v[...] = a; self.off = o + nb
        return v

    def take(self, shape, dtype):
        """An uninitialised array of this shape in the arena (256-byte aligned)."""
        dt = np.dtype(dtype); cnt = int(np.prod(shape)); nb = cnt * dt.itemsize; o = (self.off + 255) & ~255
        if o + nb > self.n:
            raise UvolError("pinned arena full")
        self.off = o + nb
        return np.frombuffer(self.buf, dtype=dt, count=cnt, offset=o).reshape(shape)

    def close(self):
        if getattr(self, "p", None):
            self.buf = None; self.L.uvol_host_free(self.p); self.p = None

    __del__ = close


def _f32(a, cols):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32).reshape(-1, cols)


def _u32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint32).reshape(-1)


class Codec:
    """One codec context = one GPU + one HIP stream.  Field names follow project-config.json."""

    def __init__(self, device=0, lib_path=None, **config):
        self.L = load(lib_path)
        p = Params()
        self.L.uvol_params_default(C.byref(p))
        for k, v in config.items():
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, int(v))
        self.params = p
        h = C.c_void_p()
        rc = self.L.uvol_ctx_create(device, C.byref(p), C.byref(h))
        if rc != UVOL_OK:
            raise UvolError(f"uvol_ctx_create(device={device}) failed rc={rc} (no CPU fallback; a gfx950 GPU is required)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.uvol_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def error(self):
        return self.L.uvol_last_error(self.h).decode()

    # ---- geometry ----
    @staticmethod
    def _mesh_host(pos, idx_pos, uv=None, idx_uv=None, nrm=None, idx_nrm=None):
        pos = _f32(pos, 3); uv = _f32(uv, 2); nrm = _f32(nrm, 3)
        idx_pos = _u32(idx_pos); idx_uv = _u32(idx_uv); idx_nrm = _u32(idx_nrm)
        m = Mesh()
        m.pos = pos.ctypes.data; m.n_pos = len(pos)
        m.idx_pos = idx_pos.ctypes.data; m.n_faces = len(idx_pos) // 3
        if uv is not None and idx_uv is not None:
            m.uv = uv.ctypes.data; m.n_uv = len(uv); m.idx_uv = idx_uv.ctypes.data
        if nrm is not None and idx_nrm is not None:
            m.nrm = nrm.ctypes.data; m.n_nrm = len(nrm); m.idx_nrm = idx_nrm.ctypes.data
        return m, (pos, uv, nrm, idx_pos, idx_uv, idx_nrm)

    def mesh_workspace(self, pos, idx_pos, uv=None, idx_uv=None, nrm=None, idx_nrm=None) -> int:
        """Device bytes one frame of these dimensions holds while in flight."""
        m, keep = self._mesh_host(pos, idx_pos, uv, idx_uv, nrm, idx_nrm)
        return int(self.L.uvol_mesh_workspace(self.h, C.byref(m)))

    def encode_mesh(self, pos, idx_pos, uv=None, idx_uv=None, nrm=None, idx_nrm=None) -> bytes:
        return self.encode_mesh_batch([dict(pos=pos, idx_pos=idx_pos, uv=uv, idx_uv=idx_uv, nrm=nrm, idx_nrm=idx_nrm)])[0]

    def encode_mesh_batch(self, frames, raise_on_error=True, views=False):
        """frames: list of dicts(pos, idx_pos[, uv, idx_uv, nrm, idx_nrm]) of host arrays -> list of .drc bytes (views=True: numpy views
        of this codec's output buffers, valid until the next call)."""
        n = len(frames)
        meshes = (Mesh * n)(); keep = []
        for i, f in enumerate(frames):
            m, k = self._mesh_host(**f); meshes[i] = m; keep.append(k)
        return self._run_batch(self.L.uvol_encode_mesh_batch, meshes, n, raise_on_error, views)

    def encode_mesh_batch_dev(self, meshes, raise_on_error=True, views=False):
        """meshes: ctypes array of Mesh holding DEVICE pointers (inputs resident in HBM).  views=True: numpy views of this
        codec's output buffers instead of `bytes` copies (valid until the next call; no 250 KB copy per frame in Python)."""
        return self._run_batch(self.L.uvol_encode_mesh_batch_dev, meshes, len(meshes), raise_on_error, views)

    def encode_mesh_batch_dev_out(self, meshes, dev_out, dev_cap, producer_stream=None):
        """GPU-resident form: `meshes` hold device pointers produced on `producer_stream` (a hipStream_t as an int, None = complete), the
        .drc bitstreams stay in HBM, packed into the caller's device buffer.  -> (offsets, lengths, statuses) as lists."""
        n = len(meshes)
        offs = (C.c_size_t * n)(); lens = (C.c_size_t * n)(); st = (C.c_int * n)()
        rc = self.L.uvol_encode_mesh_batch_dev_out(self.h, meshes, n, C.c_void_p(producer_stream or 0), C.c_void_p(int(dev_out)), int(dev_cap), offs, lens, st)
        if rc != UVOL_OK:
            raise UvolError(f"encode_mesh_batch_dev_out rc={rc}: {self.error()}")
        return list(offs), list(lens), list(st)

    def decode_mesh_batch_dev(self, files, metas):
        """files: list of .drc bytes; metas: ctypes array of DecodedMesh whose buffers are DEVICE pointers (capacities filled in by the
        caller, see uvol_drc_info): the decoded arrays stay in HBM, the counts come back in `metas`.  -> list of statuses."""
        files = [bytes(f) for f in files]; n = len(files)
        fp = (C.c_char_p * n)(*files); ln = (C.c_size_t * n)(*[len(f) for f in files]); st = (C.c_int * n)()
        rc = self.L.uvol_decode_mesh_batch_dev(self.h, fp, ln, n, metas, st)
        if rc != UVOL_OK:
            raise UvolError(f"decode_mesh_batch_dev rc={rc}: {self.error()}")
        return list(st)

    def parse_obj_batch_dev(self, texts, slot=0):
        """texts: list of OBJ files as bytes -> (ctypes array of Mesh with DEVICE pointers into the context's slot, list of statuses)."""
        texts = [bytes(t) for t in texts]; n = len(texts)
        tp = (C.c_char_p * n)(*texts); ln = (C.c_size_t * n)(*[len(t) for t in texts]); meshes = (Mesh * n)(); st = (C.c_int * n)()
        rc = self.L.uvol_parse_obj_batch_dev(self.h, tp, ln, n, slot, meshes, st)
        if rc != UVOL_OK:
            raise UvolError(f"parse_obj_batch_dev rc={rc}: {self.error()}")
        return meshes, list(st)

    def unfilter_png_batch_dev(self, inflated, width, height, channels, slot=0, sync=True):
        """inflated: list of bytes (the inflated IDAT stream of 8-bit RGB / RGBA PNGs of one size) -> list of DEVICE pointers to RGBA8 layers.
        sync=False: return once the kernel is queued (this context's texture entry points order themselves behind it)."""
        raws = [bytes(r) for r in inflated]; n = len(raws)
        rp = (C.c_char_p * n)(*raws); out = (C.c_void_p * n)()
        rc = self.L.uvol_unfilter_png_batch_dev(self.h, rp, n, width, height, channels, slot, out)
        if rc != UVOL_OK:
            raise UvolError(f"unfilter_png_batch_dev rc={rc}: {self.error()}")
        if sync and self.L.uvol_sync(self.h) != UVOL_OK:
            raise UvolError(f"uvol_sync: {self.error()}")
        return [int(p) for p in out]

    def inflate_png_batch_dev(self, zstreams, width, height, channels, slot=0, sync=True):
        """zstreams: list of bytes (the zlib stream = concatenated IDAT data of 8-bit RGB / RGBA PNGs of one size) -> (list of DEVICE pointers
        to RGBA8 layers, list of per-image statuses); inflate and un-filter both run on the device."""
        zs = [bytes(z) for z in zstreams]; n = len(zs)
        zp = (C.c_char_p * n)(*zs); ln = (C.c_size_t * n)(*[len(z) for z in zs]); out = (C.c_void_p * n)(); st = (C.c_int * n)()
        rc = self.L.uvol_inflate_png_batch_dev(self.h, zp, ln, n, width, height, channels, slot, out)
        if rc != UVOL_OK:
            raise UvolError(f"inflate_png_batch_dev rc={rc}: {self.error()}")
        rc = self.L.uvol_png_status(self.h, slot, st, n)
        if rc != UVOL_OK:
            raise UvolError(f"png_status rc={rc}: {self.error()}")
        if sync and self.L.uvol_sync(self.h) != UVOL_OK:
            raise UvolError(f"uvol_sync: {self.error()}")
        return [int(p) for p in out], list(st)

    def drc_info(self, data):
        nf, mv = C.c_uint32(), C.c_uint32()
        if self.L.uvol_drc_info(bytes(data), len(data), C.byref(nf), C.byref(mv)) != UVOL_OK:
            raise UvolError("not a .drc this decoder handles")
        return nf.value, mv.value

    # ---- enqueue form (uvol_*_async + uvol_sync): start_* record a call, finish() completes all of them in order ----
    def start_mesh_batch(self, frames, slot=None):
        """Enqueues uvol_encode_mesh_batch_async for host frames; the result arrives with finish().  slot=None: fresh output buffers, finish()
        returns `bytes`; slot=k: the output buffers of slot k are re-used by every call with that slot and finish() returns numpy views into
        them (two slots let consecutive enqueued calls overlap without a buffer set - and its page faults - per call)."""
        n = len(frames)
        meshes = (Mesh * n)(); keep = []
        for i, f in enumerate(frames):
            m, k = self._mesh_host(**f); meshes[i] = m; keep.append(k)
        caps = (C.c_size_t * n)(); lens = (C.c_size_t * n)(); st = (C.c_int * n)(); outs = (C.c_void_p * n)()
        bufs = [] if slot is None else self.__dict__.setdefault("_slot_bufs", {}).setdefault(("host", slot), [])
        while len(bufs) < n:
            bufs.append(np.empty(0, dtype=np.uint8))
        for i in range(n):
            cap = self.L.uvol_mesh_bound(C.byref(meshes[i]))
            if bufs[i].size < cap:
                bufs[i] = np.empty(cap, dtype=np.uint8)
            caps[i] = cap; outs[i] = bufs[i].ctypes.data
        rc = self.L.uvol_encode_mesh_batch_async(self.h, meshes, n, outs, caps, lens, st)
        if rc != UVOL_OK:
            raise UvolError(f"encode_mesh_batch_async rc={rc}: {self.error()}")
        self._pending = getattr(self, "_pending", []) + [("mesh" if slot is None else "mesh_views", n, bufs, lens, st, keep, meshes)]

    def start_mesh_batch_dev(self, meshes, slot=0):
        """Enqueues uvol_encode_mesh_batch_dev_async for a ctypes array of Mesh holding DEVICE pointers.  The output buffers of `slot`
        are re-used by every call with that slot (two slots let consecutive enqueued calls overlap without a buffer set per call);
        finish() returns numpy views into them."""
        n = len(meshes)
        sets = self.__dict__.setdefault("_slot_bufs", {})
        bufs = sets.setdefault(slot, [])
        while len(bufs) < n:
            bufs.append(np.empty(0, dtype=np.uint8))
        caps = (C.c_size_t * n)(); lens = (C.c_size_t * n)(); st = (C.c_int * n)(); outs = (C.c_void_p * n)()
        for i in range(n):
            cap = self.L.uvol_mesh_bound(C.byref(meshes[i]))
            if bufs[i].size < cap:
                bufs[i] = np.empty(cap, dtype=np.uint8)
            caps[i] = cap; outs[i] = bufs[i].ctypes.data
        rc = self.L.uvol_encode_mesh_batch_dev_async(self.h, meshes, n, outs, caps, lens, st)
        if rc != UVOL_OK:
            raise UvolError(f"encode_mesh_batch_dev_async rc={rc}: {self.error()}")
        self._pending = getattr(self, "_pending", []) + [("mesh_views", n, bufs, lens, st, (caps, outs), meshes)]

    def _tex_bufs(self, slot, nseg, cap):
        """Output buffers of texture slot `slot` (kept between calls: no fresh pages to fault in per batch)."""
        bufs = self.__dict__.setdefault("_slot_bufs", {}).setdefault(("tex", slot), [])
        while len(bufs) < nseg:
            bufs.append(np.empty(0, dtype=np.uint8))
        for i in range(nseg):
            if bufs[i].size < cap:
                bufs[i] = np.empty(cap, dtype=np.uint8)
        return bufs

    def start_texture_segments(self, segments, slot=None):
        """Enqueues uvol_encode_texture_segments_async for host segments (slot: as in start_mesh_batch)."""
        arrs = [[np.ascontiguousarray(a, dtype=np.uint8) for a in seg] for seg in segments]
        h, w = arrs[0][0].shape[:2]; nl = len(arrs[0]); nseg = len(arrs)
        flat = [a for seg in arrs for a in seg]
        ptrs = (C.c_void_p * len(flat))(*[a.ctypes.data for a in flat])
        cap = self.L.uvol_texture_bound(w, h, nl)
        bufs = [np.empty(cap, dtype=np.uint8) for _ in range(nseg)] if slot is None else self._tex_bufs(slot, nseg, cap)
        outs = (C.c_void_p * nseg)(*[b.ctypes.data for b in bufs[:nseg]]); caps = (C.c_size_t * nseg)(*([cap] * nseg)); lens = (C.c_size_t * nseg)()
        rc = self.L.uvol_encode_texture_segments_async(self.h, ptrs, nseg, nl, w, h, outs, caps, lens)
        if rc != UVOL_OK:
            raise UvolError(f"encode_texture_segments_async rc={rc}: {self.error()}")
        self._pending = getattr(self, "_pending", []) + [("tex" if slot is None else "tex_views", nseg, bufs, lens, None, flat, ptrs)]

    def start_texture_segments_dev(self, dev_ptrs, n_layers, width, height, slot=0):
        """Enqueues uvol_encode_texture_segments_dev_async for a flat list of n_segments * n_layers DEVICE pointers; output buffers of `slot`
        are re-used, finish() returns numpy views."""
        ptrs = (C.c_void_p * len(dev_ptrs))(*[int(p) for p in dev_ptrs]); nseg = len(dev_ptrs) // n_layers
        cap = self.L.uvol_texture_bound(width, height, n_layers)
        bufs = self._tex_bufs(("dev", slot), nseg, cap)
        outs = (C.c_void_p * nseg)(*[b.ctypes.data for b in bufs[:nseg]]); caps = (C.c_size_t * nseg)(*([cap] * nseg)); lens = (C.c_size_t * nseg)()
        rc = self.L.uvol_encode_texture_segments_dev_async(self.h, ptrs, nseg, n_layers, width, height, outs, caps, lens)
        if rc != UVOL_OK:
            raise UvolError(f"encode_texture_segments_dev_async rc={rc}: {self.error()}")
        self._pending = getattr(self, "_pending", []) + [("tex_views", nseg, bufs, lens, None, None, ptrs)]

    def trim(self):
        """uvol_trim: completes the context's work and gives its geometry workspaces back to the device (streams stay)."""
        rc = self.L.uvol_trim(self.h)
        if rc != UVOL_OK:
            raise UvolError(f"uvol_trim rc={rc}: {self.error()}")

    def finish(self):
        """uvol_sync: completes every enqueued call; returns their results in call order (meshes: None for a failed frame)."""
        rc = self.L.uvol_sync(self.h)
        pend, self._pending = getattr(self, "_pending", []), []
        res = []
        for kind, n, bufs, lens, st, _, _ in pend:
            if kind == "mesh_views":
                res.append([(bufs[i][:lens[i]] if st[i] == UVOL_OK else None) for i in range(n)])
            elif kind == "tex_views":
                res.append([(bufs[i][:lens[i]] if lens[i] else None) for i in range(n)])
            else:
                res.append([(bufs[i][:lens[i]].tobytes() if ((st is None or st[i] == UVOL_OK) and lens[i]) else None) for i in range(n)])
        if rc != UVOL_OK:
            # a call that failed as a whole leaves its lengths at zero (entries None); the calls that succeeded are not lost with it
            e = UvolError(f"uvol_sync rc={rc}: {self.error()}"); e.partial_results = res
            raise e
        return res

    def _run_batch(self, fn, meshes, n, raise_on_error, views=False):
        caps = (C.c_size_t * n)(); lens = (C.c_size_t * n)(); st = (C.c_int * n)(); outs = (C.c_void_p * n)()
        bufs = getattr(self, "_obufs", [])          # output buffers are kept between calls (no fresh pages to fault in per batch)
        while len(bufs) < n:
            bufs.append(np.empty(0, dtype=np.uint8))
        for i in range(n):
            cap = self.L.uvol_mesh_bound(C.byref(meshes[i]))
            if bufs[i].size < cap:
                bufs[i] = np.empty(cap, dtype=np.uint8)
            caps[i] = cap; outs[i] = bufs[i].ctypes.data
        self._obufs = bufs
        rc = fn(self.h, meshes, n, outs, caps, lens, st)
        if rc != UVOL_OK:
            raise UvolError(f"encode_mesh_batch rc={rc}: {self.error()}")
        res = []
        for i in range(n):
            if st[i] != UVOL_OK:
                if raise_on_error:
                    raise UvolError(f"frame {i} failed status={st[i]}: {self.error()}")
                res.append(None)
            else:
                res.append(bufs[i][:lens[i]] if views else bufs[i][:lens[i]].tobytes())
        return res

    # ---- texture ----
    def encode_texture_segment(self, layers) -> bytes:
        """layers: list of HxWx4 uint8 arrays (top row first) -> one .ktx2 (ETC1S/BasisLZ, len(layers) array layers)."""
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in layers]
        h, w = arrs[0].shape[:2]
        for a in arrs:
            if a.shape != (h, w, 4):
                raise ValueError("all layers must be HxWx4 uint8 of one size")
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        return self._run_tex(self.L.uvol_encode_texture_segment, ptrs, n, w, h)

    def encode_texture_segment_dev(self, dev_ptrs, width, height) -> bytes:
        n = len(dev_ptrs)
        ptrs = (C.c_void_p * n)(*[int(p) for p in dev_ptrs])
        return self._run_tex(self.L.uvol_encode_texture_segment_dev, ptrs, n, width, height)

    def encode_texture_segments(self, segments, views=False):
        """segments: list of lists of HxWx4 uint8 arrays (same size, same layer count) -> list of .ktx2 bytes (one batched call;
        views=True: numpy views of this codec's output buffers, valid until the next call)."""
        arrs = [[np.ascontiguousarray(a, dtype=np.uint8) for a in seg] for seg in segments]
        h, w = arrs[0][0].shape[:2]; nl = len(arrs[0])
        flat = [a for seg in arrs for a in seg]
        if any(len(seg) != nl for seg in arrs) or any(a.shape != (h, w, 4) for a in flat):
            raise ValueError("all segments must have the same layer count and HxWx4 size")
        ptrs = (C.c_void_p * len(flat))(*[a.ctypes.data for a in flat])
        return self._run_tex_batch(self.L.uvol_encode_texture_segments, ptrs, len(arrs), nl, w, h, views)

    def encode_texture_segments_dev(self, dev_ptrs, n_layers, width, height, views=False):
        """dev_ptrs: flat list of n_segments*n_layers device pointers."""
        ptrs = (C.c_void_p * len(dev_ptrs))(*[int(p) for p in dev_ptrs])
        return self._run_tex_batch(self.L.uvol_encode_texture_segments_dev, ptrs, len(dev_ptrs) // n_layers, n_layers, width, height, views)

    def encode_texture_segments_status(self, segments, caps=None):
        """Per-segment results (uvol_encode_texture_segments_st): -> (list of .ktx2 bytes or None, list of status codes).  caps: optional
        output capacities per segment (tests: a too-small one fails alone with UVOL_E_NOSPACE)."""
        arrs = [[np.ascontiguousarray(a, dtype=np.uint8) for a in seg] for seg in segments]
        h, w = arrs[0][0].shape[:2]; nl = len(arrs[0]); nseg = len(arrs)
        flat = [a for seg in arrs for a in seg]
        ptrs = (C.c_void_p * len(flat))(*[a.ctypes.data for a in flat])
        cap = self.L.uvol_texture_bound(w, h, nl)
        cl = [cap] * nseg if caps is None else [int(c) for c in caps]
        bufs = [np.empty(max(c, 1), dtype=np.uint8) for c in cl]
        outs = (C.c_void_p * nseg)(*[b.ctypes.data for b in bufs]); capa = (C.c_size_t * nseg)(*cl); lens = (C.c_size_t * nseg)(); st = (C.c_int * nseg)()
        rc = self.L.uvol_encode_texture_segments_st(self.h, ptrs, nseg, nl, w, h, 0, outs, capa, lens, st)
        if rc != UVOL_OK:
            raise UvolError(f"encode_texture_segments_st rc={rc}: {self.error()}")
        return [bufs[i][:lens[i]].tobytes() if st[i] == UVOL_OK else None for i in range(nseg)], list(st)

    TARGETS = {"rgba32": (0, 4, False), "etc1": (1, 8, True), "bc7": (2, 16, True), "astc": (3, 16, True), "etc2_rgba": (4, 16, True), "bc1": (5, 8, True), "bc3": (6, 16, True)}

    def transcode_texture_segments_status(self, files, target="rgba32", shape=None):
        """Per-segment results and mixed batches (uvol_transcode_texture_segments_st): files may mix ETC1S and UASTC sources and contain
        unreadable or corrupt ones -> (list of arrays or None, list of status codes).  shape = (w, h, layers) if the first file is not readable."""
        files = [bytes(f) for f in files]; n = len(files)
        code, unit, blocks = self.TARGETS[target]
        if shape is None:
            for f in files:
                try:
                    shape = self.ktx2_info(f); break
                except UvolError:
                    continue
        w, h, nl = shape; bx, by = (w + 3) // 4, (h + 3) // 4
        outs = [np.zeros((nl, by, bx, unit) if blocks else (nl, h, w, 4), dtype=np.uint8) for _ in range(n)]
        fp = (C.c_char_p * n)(*files); ln = (C.c_size_t * n)(*[len(f) for f in files]); st = (C.c_int * n)()
        ptrs = (C.c_void_p * (n * nl))(*[outs[s][l].ctypes.data for s in range(n) for l in range(nl)])
        rc = self.L.uvol_transcode_texture_segments_st(self.h, fp, ln, n, ptrs, bx * by * unit if blocks else w * h * 4, 0, code, st)
        if rc != UVOL_OK:
            raise UvolError(f"transcode_texture_segments_st rc={rc}: {self.error()}")
        return [outs[i] if st[i] == UVOL_OK else None for i in range(n)], list(st)

    def _run_tex_batch(self, fn, ptrs, nseg, nl, w, h, views=False):
        cap = self.L.uvol_texture_bound(w, h, nl)
        bufs = self._tex_bufs("blocking", nseg, cap)
        outs = (C.c_void_p * nseg)(*[b.ctypes.data for b in bufs[:nseg]]); caps = (C.c_size_t * nseg)(*([cap] * nseg)); lens = (C.c_size_t * nseg)()
        rc = fn(self.h, ptrs, nseg, nl, w, h, outs, caps, lens)
        if rc != UVOL_OK:
            raise UvolError(f"encode_texture_segments rc={rc}: {self.error()}")
        return [(bufs[i][:lens[i]] if views else bufs[i][:lens[i]].tobytes()) for i in range(nseg)]

    def _run_tex(self, fn, ptrs, n, w, h):
        cap = self.L.uvol_texture_bound(w, h, n)
        out = np.empty(cap, dtype=np.uint8); ln = C.c_size_t()
        rc = fn(self.h, ptrs, n, w, h, out.ctypes.data, cap, C.byref(ln))
        if rc != UVOL_OK:
            raise UvolError(f"encode_texture_segment rc={rc}: {self.error()}")
        return out[:ln.value].tobytes()

    # ---- decode path (texture half) ----
    def ktx2_info(self, data: bytes):
        w, h, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = self.L.uvol_ktx2_info(data, len(data), C.byref(w), C.byref(h), C.byref(n))
        if rc != UVOL_OK:
            raise UvolError(f"uvol_ktx2_info rc={rc}")
        return w.value, h.value, n.value

    def decode_texture_segments(self, files, out=None):
        """files: list of .ktx2 bytes (one width / height / layer count) -> list (per segment) of [layers, H, W, 4] uint8 arrays,
        rows in stored order (the encoder's -y_flip is part of the stored image)."""
        files = [bytes(f) for f in files]
        w, h, nl = self.ktx2_info(files[0]); n = len(files)
        outs = out if out is not None else [np.empty((nl, h, w, 4), dtype=np.uint8) for _ in range(n)]      # out: arrays of an earlier call, re-used
        fp = (C.c_char_p * n)(*files); ln = (C.c_size_t * n)(*[len(f) for f in files])
        ptrs = (C.c_void_p * (n * nl))(*[outs[s][l].ctypes.data for s in range(n) for l in range(nl)])
        rc = self.L.uvol_decode_texture_segments(self.h, fp, ln, n, ptrs, w * h * 4)
        if rc != UVOL_OK:
            raise UvolError(f"decode_texture_segments rc={rc}: {self.error()}")
        return outs

    def transcode_texture_segments_etc1(self, files):
        """files: list of .ktx2 bytes -> list (per segment) of [layers, by, bx, 8] uint8 arrays of ETC1 blocks (raster order)."""
        files = [bytes(f) for f in files]
        w, h, nl = self.ktx2_info(files[0]); n = len(files); bx, by = (w + 3) // 4, (h + 3) // 4
        outs = [np.empty((nl, by, bx, 8), dtype=np.uint8) for _ in range(n)]
        fp = (C.c_char_p * n)(*files); ln = (C.c_size_t * n)(*[len(f) for f in files])
        ptrs = (C.c_void_p * (n * nl))(*[outs[s][l].ctypes.data for s in range(n) for l in range(nl)])
        rc = self.L.uvol_transcode_texture_segments_etc1(self.h, fp, ln, n, ptrs, bx * by * 8, 0)
        if rc != UVOL_OK:
            raise UvolError(f"transcode_texture_segments_etc1 rc={rc}: {self.error()}")
        return outs

    def transcode_texture_segments_bc7(self, files):
        """files: list of .ktx2 bytes -> list (per segment) of [layers, by, bx, 16] uint8 arrays of BC7 mode-6 blocks (raster order)."""
        files = [bytes(f) for f in files]
        w, h, nl = self.ktx2_info(files[0]); n = len(files); bx, by = (w + 3) // 4, (h + 3) // 4
        outs = [np.empty((nl, by, bx, 16), dtype=np.uint8) for _ in range(n)]
        fp = (C.c_char_p * n)(*files); ln = (C.c_size_t * n)(*[len(f) for f in files])
        ptrs = (C.c_void_p * (n * nl))(*[outs[s][l].ctypes.data for s in range(n) for l in range(nl)])
        rc = self.L.uvol_transcode_texture_segments_bc7(self.h, fp, ln, n, ptrs, bx * by * 16, 0)
        if rc != UVOL_OK:
            raise UvolError(f"transcode_texture_segments_bc7 rc={rc}: {self.error()}")
        return outs

    def transcode_texture_segments_etc2_rgba(self, files):
        """files: list of .ktx2 bytes (with or without alpha slices) -> list (per segment) of [layers, by, bx, 16] uint8 arrays of ETC2 RGBA blocks."""
        files = [bytes(f) for f in files]
        w, h, nl = self.ktx2_info(files[0]); n = len(files); bx, by = (w + 3) // 4, (h + 3) // 4
        outs = [np.empty((nl, by, bx, 16), dtype=np.uint8) for _ in range(n)]
        fp = (C.c_char_p * n)(*files); ln = (C.c_size_t * n)(*[len(f) for f in files])
        ptrs = (C.c_void_p * (n * nl))(*[outs[s][l].ctypes.data for s in range(n) for l in range(nl)])
        rc = self.L.uvol_transcode_texture_segments_etc2_rgba(self.h, fp, ln, n, ptrs, bx * by * 16, 0)
        if rc != UVOL_OK:
            raise UvolError(f"transcode_texture_segments_etc2_rgba rc={rc}: {self.error()}")
        return outs

    def transcode_texture_segments_astc(self, files):
        """files: list of UASTC .ktx2 bytes -> list (per segment) of [layers, by, bx, 16] uint8 arrays of ASTC 4x4 blocks (raster order)."""
        files = [bytes(f) for f in files]
        w, h, nl = self.ktx2_info(files[0]); n = len(files); bx, by = (w + 3) // 4, (h + 3) // 4
        outs = [np.empty((nl, by, bx, 16), dtype=np.uint8) for _ in range(n)]
        fp = (C.c_char_p * n)(*files); ln = (C.c_size_t * n)(*[len(f) for f in files])
        ptrs = (C.c_void_p * (n * nl))(*[outs[s][l].ctypes.data for s in range(n) for l in range(nl)])
        rc = self.L.uvol_transcode_texture_segments_astc(self.h, fp, ln, n, ptrs, bx * by * 16, 0)
        if rc != UVOL_OK:
            raise UvolError(f"transcode_texture_segments_astc rc={rc}: {self.error()}")
        return outs

    def decode_texture_segments_dev(self, files, dev_ptrs, layer_cap):
        """Same, into caller-owned device buffers: dev_ptrs = flat list of n_segments*layers device pointers."""
        files = [bytes(f) for f in files]; n = len(files)
        fp = (C.c_char_p * n)(*files); ln = (C.c_size_t * n)(*[len(f) for f in files])
        ptrs = (C.c_void_p * len(dev_ptrs))(*[int(p) for p in dev_ptrs])
        rc = self.L.uvol_decode_texture_segments_dev(self.h, fp, ln, n, ptrs, layer_cap)
        if rc != UVOL_OK:
            raise UvolError(f"decode_texture_segments_dev rc={rc}: {self.error()}")

    # ---- decode path (geometry half) ----
    def decode_arena_bytes(self, files):
        """Bytes of a PinnedArena that holds the output arrays of decode_mesh_batch(files, arena=...): capacities by uvol_drc_info (the counts
        are not known before the decode: 3 x faces values per attribute at most)."""
        tot = 0
        for f in files:
            nf, mv = C.c_uint32(), C.c_uint32()
            if self.L.uvol_drc_info(bytes(f), len(f), C.byref(nf), C.byref(mv)) != UVOL_OK:
                raise UvolError("not a .drc this decoder handles")
            tot += mv.value * 32 + 3 * nf.value * 12 + 6 * 256
        return tot + 4096

    def decode_mesh_batch(self, files, raise_on_error=True, fetch=True, views=False, arena=None):
        """files: list of .drc bytes -> list of dicts {pos [n,3], uv [n,2], nrm [n,3] float32 in decoding order,
        idx_pos / idx_uv / idx_nrm [3*faces] uint32 entry index per corner}; absent attributes are None.
        arena (a PinnedArena, with views=True): the output arrays are carved from it - outputs that all lie in uvol_host_alloc memory are written
        by the DMA engines where they are, without the library's staging buffers."""
        files = [bytes(f) for f in files]; n = len(files)
        metas = (DecodedMesh * n)(); keep = []
        pool = self.__dict__.setdefault("_dec_bufs_pinned" if arena is not None else "_dec_bufs", []) if views else None      # views=True: the arrays are kept and re-used by the next call
        mk = (lambda shape, dt: arena.take(shape, dt)) if arena is not None else (lambda shape, dt: np.empty(shape, dt))
        for i, f in enumerate(files):
            nf, mv = C.c_uint32(), C.c_uint32()
            if self.L.uvol_drc_info(f, len(f), C.byref(nf), C.byref(mv)) != UVOL_OK:
                raise UvolError(f"frame {i}: not a .drc this decoder handles")
            a = pool[i] if (pool is not None and i < len(pool) and pool[i]["idx_pos"].size >= 3 * nf.value and pool[i]["pos"].shape[0] >= mv.value) else None
            if a is None:
                a = dict(pos=mk((mv.value, 3), np.float32), uv=mk((mv.value, 2), np.float32), nrm=mk((mv.value, 3), np.float32),
                         idx_pos=mk((3 * nf.value,), np.uint32), idx_uv=mk((3 * nf.value,), np.uint32), idx_nrm=mk((3 * nf.value,), np.uint32))
                if pool is not None:
                    if i < len(pool):
                        pool[i] = a
                    else:
                        pool.append(a)
            keep.append(a)
            m = metas[i]; m.cap_faces = nf.value; m.cap_values = mv.value
            for k, v in a.items():
                setattr(m, k, v.ctypes.data if fetch else None)        # fetch=False: decode only, results stay on the device (timing)
        fp = (C.c_char_p * n)(*files); ln = (C.c_size_t * n)(*[len(f) for f in files]); st = (C.c_int * n)()
        rc = self.L.uvol_decode_mesh_batch(self.h, fp, ln, n, metas, st)
        if rc != UVOL_OK:
            raise UvolError(f"decode_mesh_batch rc={rc}: {self.error()}")
        res = []
        for i in range(n):
            if st[i] != UVOL_OK:
                if raise_on_error:
                    raise UvolError(f"frame {i} failed status={st[i]}: {self.error()}")
                res.append(None); continue
            m, a = metas[i], keep[i]
            cnt = dict(pos=m.n_pos, uv=m.n_uv, nrm=m.n_nrm)
            if not fetch:
                res.append(dict(n_faces=m.n_faces, n_pos=m.n_pos, n_uv=m.n_uv, n_nrm=m.n_nrm)); continue
            cp = (lambda v: v) if views else (lambda v: v.copy())
            res.append({k: (cp(a[k][:cnt[k]]) if cnt[k] else None) for k in ("pos", "uv", "nrm")} |
                       {"idx_" + k: (cp(a["idx_" + k][:3 * m.n_faces]) if cnt[k] else None) for k in ("pos", "uv", "nrm")} | {"n_faces": m.n_faces})
        return res

    # ---- measurement ----
    def profile(self, on=True):
        self.L.uvol_profile_enable(self.h, 1 if on else 0)

    def profile_reset(self):
        self.L.uvol_profile_reset(self.h)

    def profile_report(self):
        self.L.uvol_sync(self.h)
        out = []
        for i in range(self.L.uvol_profile_count(self.h)):
            name = C.create_string_buffer(128); la = C.c_uint64(); ms = C.c_double(); ab = C.c_uint64()
            self.L.uvol_profile_get(self.h, i, name, 128, C.byref(la), C.byref(ms), C.byref(ab))
            out.append(dict(name=name.value.decode(), launches=la.value, total_ms=ms.value, algo_bytes=ab.value))
        return out
