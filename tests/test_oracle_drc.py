"""CPU oracle vs the reference's own fixtures + oracle encoder round trips."""
import json
import os
import zlib
import numpy as np
import pytest
from conftest import GOLDEN, REF_OUT
from helpers import check_roundtrip

GOLD = json.load(open(os.path.join(GOLDEN, "drc_goldens.json")))


def _check_file(O, data, g):
    m = O.drc_decode(data)
    p, u, n = m.att("position"), m.att("tex_coord"), m.att("normal")
    assert len(data) == g["size"] and m.leftover == 0
    assert (m.nev, m.nf, m.nad, m.nsym, m.nsplit, m.nts) == (g["nev"], g["nf"], g["nad"], g["nsym"], g["nsplit"], g["nts"])
    assert m.ctx_n == g["ctx_n"] and (m.conn_end, m.hdr_end) == (g["conn_end"], g["hdr_end"])
    assert (p["n"], u["n"], n["n"], u["n_orient"], n["n_flip_set"]) == (g["n_pos"], g["n_uv"], g["n_nrm"], g["n_orient"], g["n_flip"])
    assert "%08x" % O.crc32(p["vals"].astype(np.int32)) == g["crc_pos"]
    assert "%08x" % O.crc32(u["vals"].astype(np.int32)) == g["crc_uv"]
    assert "%08x" % O.crc32(n["vals"].astype(np.int32)) == g["crc_nrm"]
    assert "%08x" % O.crc32(m.c2v.astype(np.int32)) == g["crc_c2v"]
    # the encoder settings of scripts/Encoder.py:260 (-qp 11 -qt 10 -qn 8 -cl 7)
    assert (p["pred_method"], p["transform"], p["qbits"]) == (1, 1, 11)
    assert (u["pred_method"], u["transform"], u["qbits"]) == (5, 1, 10)
    assert (n["pred_method"], n["transform"], n["qbits"]) == (6, 3, 8)
    return m


def test_survey_appendix_c_spot_values():
    """Numbers quoted in SURVEY.md Appendix C (independent of this repo's generator)."""
    f0 = GOLD["files"]["00000.drc"]
    assert (f0["size"], f0["nev"], f0["nf"], f0["nsym"], f0["nsplit"], f0["nts"]) == (95543, 26145, 52290, 52289, 2389, 2)
    assert f0["ctx_n"] == [3088, 3165, 8442, 14617, 13727, 9249]
    assert (f0["crc_pos"], f0["crc_uv"], f0["crc_nrm"], f0["crc_c2v"]) == ("1b1cd349", "7865e2db", "673fe693", "52e65602")
    assert (f0["sum_pos"], f0["sum_uv"], f0["sum_nrm"]) == (49231631, 34790842, 6214557)
    assert GOLD["aggregate_crc"] == "38cada54" and GOLD["n_files"] == 250
    s = GOLD["files"]
    assert sum(s[k]["nev"] for k in s) == 6835821 and sum(s[k]["nf"] for k in s) == 13671038
    assert sum(s[k]["n_uv"] for k in s) == 8375918 and sum(s[k]["n_nrm"] for k in s) == 6838937
    assert sum(s[k]["n_orient"] for k in s) == 8068841
    assert (s["00075.drc"]["crc_pos"], s["00249.drc"]["crc_uv"]) == ("744ef362", "2872aa48")


@pytest.mark.parametrize("name", ["00000.drc", "00075.drc"])
def test_decoder_on_committed_fixtures(oracle, name):
    _check_file(oracle, open(os.path.join(GOLDEN, name), "rb").read(), GOLD["files"][name])


@pytest.mark.skipif(not os.path.isdir(REF_OUT), reason="/root/reference only exists in the build container")
def test_decoder_on_all_250_reference_fixtures(oracle):
    agg = ""
    for name in sorted(GOLD["files"]):
        g = GOLD["files"][name]
        _check_file(oracle, open(os.path.join(REF_OUT, "geometry_draco", name), "rb").read(), g)
        agg += g["crc_pos"] + g["crc_uv"]
    assert "%08x" % zlib.crc32(agg.encode()) == "38cada54"


@pytest.mark.parametrize("name", ["00000.drc", "00075.drc"])
def test_entropy_coder_kat_byte_identical(oracle, name):
    """decode each rANS section of a stock draco_encoder file -> re-encode -> identical bytes."""
    b = open(os.path.join(GOLDEN, name), "rb").read()
    m = oracle.drc_decode(b)
    for a in m.atts:
        syms, end, info = oracle.decode_symbols(b, a["sym_begin"], a["n"] * a["ncomp_port"])
        assert end == a["sym_end"] and info.left == 0 and info.final_state == info.base
        assert oracle.encode_symbols(syms) == b[a["sym_begin"]:a["sym_end"]], a["name"]


def test_rabs_roundtrip(oracle):
    rng = np.random.default_rng(3)
    for n, p in [(0, 0.5), (1, 1.0), (7, 0.1), (1000, 0.03), (5000, 0.5), (4096, 0.999)]:
        bits = (rng.random(n) < p).astype(np.uint8)
        enc = oracle.rabs_encode(bits)
        import ctypes as C
        r = oracle.lib()
        class RD(C.Structure):
            _fields_ = [("buf", C.c_void_p), ("off", C.c_size_t), ("st", C.c_uint32), ("p0", C.c_uint8), ("end", C.c_size_t)]
        rd = RD()
        r.orc_rabs_open.argtypes = [C.POINTER(RD), C.c_char_p, C.c_size_t, C.c_size_t]
        r.orc_rabs_bit.argtypes = [C.POINTER(RD)]
        assert r.orc_rabs_open(C.byref(rd), enc, len(enc), 0) == 0
        got = [r.orc_rabs_bit(C.byref(rd)) for _ in range(n)]
        assert got == list(bits) and rd.end == len(enc)


def _meshes():
    import synth
    return {"sphere": synth.sphere_mesh(40, 21, charts=(5, 4)), "grid_with_hole": synth.grid_mesh(),
            "torus": synth.torus_mesh(), "sphere_nocrease": synth.sphere_mesh(24, 13, charts=(3, 2), crease=False)}


@pytest.mark.parametrize("name", ["sphere", "grid_with_hole", "torus", "sphere_nocrease"])
def test_encoder_roundtrip_synthetic(oracle, name):
    m = _meshes()[name]
    e = oracle.drc_encode(m["pos"], m["idx_pos"], m["uv"], m["idx_uv"], m["nrm"], m["idx_nrm"])
    d = check_roundtrip(oracle, m, e)
    assert d.nad == 2


def test_encoder_optional_attributes_and_degenerates(oracle):
    import synth
    m = synth.torus_mesh(16, 8)
    e = oracle.drc_encode(m["pos"], m["idx_pos"])
    d = check_roundtrip(oracle, dict(pos=m["pos"], idx_pos=m["idx_pos"]), e)
    assert d.nad == 0 and len(d.atts) == 1
    # duplicate a vertex value and add a degenerate face: dedup + drop must keep the result identical
    pos = np.concatenate([m["pos"], m["pos"][:1]]); idx = np.concatenate([m["idx_pos"], np.array([0, len(m["pos"]), 5], np.uint32)])
    e2 = oracle.drc_encode(pos, idx)
    assert e2 == e


@pytest.mark.parametrize("name", ["00000.drc", "00075.drc"])
def test_encoder_on_reference_geometry(oracle, name):
    """Re-encode a decoded reference frame (positions nudged inside their quantisation cell so that
    values stay distinct): same triangles back, size within 1 % of the stock draco_encoder output."""
    b = open(os.path.join(GOLDEN, name), "rb").read()
    m = oracle.drc_decode(b)
    p, u, n = m.att("position"), m.att("tex_coord"), m.att("normal")
    i = np.arange(p["n"])
    pos = p["float"] + np.stack([i % 64, (i // 64) % 64, i // 4096], 1).astype(np.float32) * np.float32(0.004)
    mesh = dict(pos=pos, idx_pos=p["corner_to_entry"], uv=u["float"], idx_uv=u["corner_to_entry"], nrm=n["float"], idx_nrm=n["corner_to_entry"])
    e = oracle.drc_encode(**mesh)
    d = check_roundtrip(oracle, mesh, e)
    assert (d.nf, d.nev) == (m.nf, m.nev)
    assert abs(len(e) - len(b)) < 0.01 * len(b)


def test_encoder_edge_case_meshes(oracle):
    """Non-manifold edges / vertices, flipped and duplicate faces, several components, open boundary: valid streams that decode to the input."""
    import synth
    for name, m in synth.edge_case_meshes().items():
        e = oracle.drc_encode(m["pos"], m["idx_pos"], m.get("uv"), m.get("idx_uv"), m.get("nrm"), m.get("idx_nrm"))
        check_roundtrip(oracle, m, e)


@pytest.mark.parametrize("qp,qt,qn", [(14, 12, 10), (8, 8, 6), (16, 16, 12)])
def test_encoder_other_quantisation_bits(oracle, qp, qt, qn):
    import synth
    m = synth.torus_mesh(16, 8)
    e = oracle.drc_encode(m["pos"], m["idx_pos"], m["uv"], m["idx_uv"], m["nrm"], m["idx_nrm"], qp=qp, qt=qt, qn=qn)
    d = check_roundtrip(oracle, m, e, qp=qp, qt=qt)
    assert [a["qbits"] for a in d.atts] == [qp, qt, qn]


def test_encoder_tool_set_options_round_trip(oracle):
    """The restatement's other tool sets - edgebreaker with the STANDARD traversal (method 1) and SEQUENTIAL connectivity with the
    difference predictor (method 2) - have no reference fixture; they are pinned by the round trip through the decoder:
    same triangles, positions within half a step; the standard traversal decodes to the very arrays of the valence stream."""
    import synth
    from helpers import check_roundtrip
    for m in (synth.torus_mesh(), synth.grid_mesh(), synth.sphere_mesh(60, 31, charts=(6, 5))):
        a = oracle.drc_encode(m["pos"], m["idx_pos"], m["uv"], m["idx_uv"], m["nrm"], m["idx_nrm"])
        b = oracle.drc_encode(m["pos"], m["idx_pos"], m["uv"], m["idx_uv"], m["nrm"], m["idx_nrm"], method=1)
        c = oracle.drc_encode(m["pos"], m["idx_pos"], m["uv"], m["idx_uv"], m["nrm"], m["idx_nrm"], method=2)
        da, db, dc = check_roundtrip(oracle, m, a), check_roundtrip(oracle, m, b), check_roundtrip(oracle, m, c)
        assert (db.method, db.traversal, dc.method) == (1, 0, 0) and dc.opp is None
        for x, y in zip(da.atts, db.atts):
            assert np.array_equal(x["vals"], y["vals"]) and np.array_equal(x["corner_to_entry"], y["corner_to_entry"])
        assert all(a_["n"] == dc.npoints for a_ in dc.atts)
