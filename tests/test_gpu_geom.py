"""Parity tests proper: the HIP path on a real MI355X, through the C-ABI, against the CPU oracle."""
import os
import numpy as np
import pytest
from conftest import GOLDEN, ROOT
from helpers import check_roundtrip

pytestmark = pytest.mark.gpu


def _small():
    import synth
    return [synth.sphere_mesh(40, 21, charts=(5, 4)), synth.grid_mesh(), synth.torus_mesh(),
            synth.sphere_mesh(24, 13, charts=(3, 2), crease=False), synth.sphere_mesh(120, 61, charts=(12, 6), frame=3)]


def _oracle_bytes(O, f):
    return O.drc_encode(f["pos"], f["idx_pos"], f.get("uv"), f.get("idx_uv"), f.get("nrm"), f.get("idx_nrm"))


def test_gpu_small_batch_bit_exact(oracle, gpu_codec):
    frames = _small()
    t = frames[2]
    frames.append(dict(pos=t["pos"], idx_pos=t["idx_pos"]))
    frames.append(dict(pos=np.concatenate([t["pos"], t["pos"][:1]]), idx_pos=np.concatenate([t["idx_pos"], np.array([0, len(t["pos"]), 5], np.uint32)])))
    res = gpu_codec.encode_mesh_batch(frames)
    for f, r in zip(frames, res):
        assert r == _oracle_bytes(oracle, f)


def test_gpu_50k_vertex_frame_bit_exact_and_index_arrays(oracle, gpu_codec):
    """BASELINE.json configs[1]: single 50k-vert frame (quantize+edgebreaker+rANS), index-array bit-exact check."""
    import synth
    m = synth.sphere_mesh(283, 177)
    assert 49000 < len(m["pos"]) < 51000
    r = gpu_codec.encode_mesh(**m)
    assert r == _oracle_bytes(oracle, m)
    check_roundtrip(oracle, m, r)


def test_gpu_reference_geometry_reencode(oracle, gpu_codec):
    """A real captured frame (decoded reference fixture) through the HIP path: byte-exact vs oracle, triangles preserved."""
    b = open(os.path.join(GOLDEN, "00000.drc"), "rb").read()
    m = oracle.drc_decode(b)
    p, u, n = m.att("position"), m.att("tex_coord"), m.att("normal")
    i = np.arange(p["n"])
    pos = p["float"] + np.stack([i % 64, (i // 64) % 64, i // 4096], 1).astype(np.float32) * np.float32(0.004)
    mesh = dict(pos=pos, idx_pos=p["corner_to_entry"], uv=u["float"], idx_uv=u["corner_to_entry"], nrm=n["float"], idx_nrm=n["corner_to_entry"])
    r = gpu_codec.encode_mesh(**mesh)
    assert r == _oracle_bytes(oracle, mesh)
    d = check_roundtrip(oracle, mesh, r)
    assert (d.nf, d.nev) == (m.nf, m.nev) and abs(len(r) - len(b)) < 0.01 * len(b)


def test_gpu_full_size_batch_properties(oracle, gpu_codec):
    """100k-vertex frames (BASELINE headline shape), a batch in flight: decode -> same triangles, positions within half a step."""
    import synth
    frames = [synth.sphere_mesh(frame=k) for k in range(3)]
    assert len(frames[0]["pos"]) == 100002 and len(frames[0]["idx_pos"]) == 600000
    res = gpu_codec.encode_mesh_batch(frames)
    for f, r in zip(frames, res):
        check_roundtrip(oracle, f, r)
    assert res[0] == _oracle_bytes(oracle, frames[0])


def test_gpu_error_isolation(gpu_codec):
    pos = np.zeros((3, 3), np.float32); pos[1, 0] = 1; pos[2, 1] = 1
    good = dict(pos=pos, idx_pos=np.array([0, 1, 2], np.uint32))
    bad = dict(pos=pos, idx_pos=np.array([0, 1, 7], np.uint32))
    res = gpu_codec.encode_mesh_batch([bad, good], raise_on_error=False)
    assert res[0] is None and res[1] is not None and res[1][:5] == b"DRACO"


def test_gpu_edge_cases_and_quantisation_bits(oracle, gpu_codec):
    import synth, uvol
    cases = list(synth.edge_case_meshes().values())
    for f, r in zip(cases, gpu_codec.encode_mesh_batch(cases)):
        assert r == _oracle_bytes(oracle, f)
    m = synth.torus_mesh(16, 8)
    for qp, qt, qn in [(14, 12, 10), (8, 8, 6), (16, 16, 12)]:
        c2 = uvol.Codec(device=0, Q_POSITION_ATTR=qp, Q_TEXTURE_ATTR=qt, Q_NORMAL_ATTR=qn)
        assert c2.encode_mesh(**m) == oracle.drc_encode(m["pos"], m["idx_pos"], m["uv"], m["idx_uv"], m["nrm"], m["idx_nrm"], qp=qp, qt=qt, qn=qn)
        c2.close()


def test_gpu_walker_bitmap_placements(oracle):
    """The serial walkers keep their visited bitmaps in LDS, fall back to a global vertex bitmap when a table has more
    vertices than the LDS slot holds, and to global memory altogether for meshes too large for LDS: all bit-exact.
    (UVOL_WALK_FORCE is read once per process, so each placement runs in its own interpreter.)"""
    import subprocess, sys, os
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import synth, uvol\nimport oracle as o\n"
        "o.lib(); c = uvol.Codec(device=0)\n"
        "frames = [synth.torus_mesh(16, 8), synth.sphere_mesh(120, 61, charts=(12, 6), frame=3), synth.grid_mesh()]\n"
        "for f, r in zip(frames, c.encode_mesh_batch(frames)):\n"
        "    assert r == o.drc_encode(f['pos'], f['idx_pos'], f.get('uv'), f.get('idx_uv'), f.get('nrm'), f.get('idx_nrm'))\n"
        "print('ok')\n"
    ) % (os.path.join(ROOT, "universal-volumetric_amd"), os.path.join(ROOT, "oracle"))
    # rec16: 16-byte corner records (meshes with >= 2^18 faces); simtN: lane-per-walker kernels, N lanes per wave (large batches);
    # entwave: the wave-per-stream entropy coder instead of the lane-per-stream one
    # relabel: the locality relabelling forced on for these coherently stored meshes (per frame it is decided on the device)
    # earlyjoin: the auxiliary stream joined before the record tables (batches above 1200 frames); small batches join late
    for force in ("", "vglobal", "global", "rec16", "simt4", "simt64", "simtcorner", "simtrec16", "entwave", "relabel", "relabel_simt", "earlyjoin"):
        env = dict(os.environ, UVOL_SIMT_W="5", UVOL_REC_FACE="0") if force == "simtcorner" else dict(os.environ, UVOL_SIMT_W="3", UVOL_REC16="1") if force == "simtrec16" else dict(os.environ, UVOL_LATE_JOIN="0") if force == "earlyjoin" else dict(os.environ, UVOL_RELABEL="1") if force == "relabel" else dict(os.environ, UVOL_RELABEL="1", UVOL_SIMT_W="5") if force == "relabel_simt" else dict(os.environ, UVOL_REC16="1") if force == "rec16" else (dict(os.environ, UVOL_SIMT_W=force[4:], UVOL_ENTROPY_W="64") if force.startswith("simt") else (dict(os.environ, UVOL_ENTROPY_WAVE="1") if force == "entwave" else dict(os.environ, UVOL_WALK_FORCE=force)))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, (force, r.stdout[-500:], r.stderr[-1500:])


def test_gpu_mesh_decode_matches_oracle(oracle, gpu_codec):
    """Decode path, geometry half on the GPU vs the pinned oracle decoder: the reference's own .drc fixtures
    (Draco 2.2, 26k vertices, 4 attribute decoders) and this codec's output for several topologies, in one batch."""
    from test_hipemu_geom import _check_decoded
    frames = _small()
    t = frames[2]
    frames.append(dict(pos=t["pos"], idx_pos=t["idx_pos"]))
    files = gpu_codec.encode_mesh_batch(frames)
    # 16-bit quantisation of every attribute (the largest operands the tex-coord predictor's f64 form must keep exact), 4 bits, and
    # a 50k-vertex frame at 16 bits: long chains of entries that refer to each other inside a 64-entry chunk
    import synth
    big = synth.sphere_mesh(280, 180, frame=1)
    for f, qb in ((frames[0], 16), (frames[1], 16), (frames[0], 4), (big, 16), (big, 11)):
        files.append(oracle.drc_encode(f["pos"], f["idx_pos"], f.get("uv"), f.get("idx_uv"), f.get("nrm"), f.get("idx_nrm"), qp=qb, qt=qb, qn=qb))
    files += [open(os.path.join(GOLDEN, n), "rb").read() for n in ("00000.drc", "00075.drc")]
    # a header whose vertex count is off by one: the symbol streams decoded beside the traversal with that count are decoded again
    lie = bytearray(files[0])
    if lie[8] == 1 and lie[12] & 0x7f < 127:
        lie[12] += 1
        files.append(bytes(lie))
    for data, got in zip(files, gpu_codec.decode_mesh_batch(files)):
        _check_decoded(oracle, data, got)


def test_gpu_traverse_vbits_l2_at_bench_size(oracle, gpu_codec):
    """bench.py's default contexts keep the attribute traversers' vertex bitmaps in L2 (`traverse_vbits_l2`, workgroup-scope
    loads next to atomic ORs): bit-identical to the LDS placement and to the oracle on 100k-vertex frames, several frames per
    batch so that many traversers share an XCD's L2."""
    import synth, uvol
    frames = [synth.sphere_mesh(frame=k) for k in range(6)]
    c2 = uvol.Codec(device=0, traverse_vbits_l2=1)
    try:
        got = c2.encode_mesh_batch(frames)
    finally:
        c2.close()
    assert got == gpu_codec.encode_mesh_batch(frames)
    f = frames[0]
    assert got[0] == oracle.drc_encode(f["pos"], f["idx_pos"], f.get("uv"), f.get("idx_uv"), f.get("nrm"), f.get("idx_nrm"))


def test_gpu_mesh_roundtrip_at_bench_size(oracle, gpu_codec):
    """encode -> decode of a 100k-vertex / 200k-face frame: identical with the oracle decoder, and order-independent
    properties of the round trip: same bounding box within a quantisation step, same total surface area within 2 %,
    unit normals, finite uvs inside the input's uv range."""
    import synth
    from test_hipemu_geom import _check_decoded
    m = synth.sphere_mesh(frame=1)
    data = gpu_codec.encode_mesh(**m)
    d = gpu_codec.decode_mesh_batch([data])[0]
    _check_decoded(oracle, data, d)
    # round 6: the same frame (x 40: a staged download would need its pinned buffers) with the output arrays in uvol_host_alloc memory -
    # written by the DMA engines where they lie, equal to the staged results
    import uvol
    ar = uvol.PinnedArena(gpu_codec.decode_arena_bytes([data] * 40))
    try:
        for dp in gpu_codec.decode_mesh_batch([data] * 40, views=True, arena=ar):
            assert all(np.array_equal(dp[k], d[k]) for k in ("pos", "uv", "nrm", "idx_pos", "idx_uv", "idx_nrm"))
    finally:
        gpu_codec.__dict__.pop("_dec_bufs_pinned", None); ar.close()
    nf = len(m["idx_pos"]) // 3
    assert d["n_faces"] == nf
    step = float((m["pos"].max(0) - m["pos"].min(0)).max()) / (2 ** 11 - 1)
    assert np.abs(d["pos"].min(0) - m["pos"].min(0)).max() <= step and np.abs(d["pos"].max(0) - m["pos"].max(0)).max() <= step

    def area(pos, idx):
        t = pos[idx].reshape(-1, 3, 3).astype(np.float64)
        return 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1).sum()
    a_in, a_out = area(m["pos"], m["idx_pos"]), area(d["pos"], d["idx_pos"])
    assert abs(a_out - a_in) <= 0.02 * a_in, (a_in, a_out)      # quantisation noise inflates the area of a dense mesh slightly
    assert np.allclose(np.linalg.norm(d["nrm"], axis=1), 1.0, atol=1e-3)
    uv_step = float((m["uv"].max(0) - m["uv"].min(0)).max()) / (2 ** 10 - 1)
    assert (d["uv"] >= m["uv"].min(0) - uv_step).all() and (d["uv"] <= m["uv"].max(0) + uv_step).all()


def test_gpu_baseline_config1_single_50k_vertex_frame(oracle, gpu_codec):
    """BASELINE.json configs[1]: one 50k-vertex frame through quantise + edgebreaker + rANS on the GPU; the decoded
    triangle index arrays equal the input after canonicalisation (helpers.check_roundtrip) and the bytes equal the oracle's;
    then the same frame back through the GPU decoder."""
    import synth
    from test_hipemu_geom import _check_decoded
    m = synth.sphere_mesh(283, 177, charts=(28, 18), frame=2)
    assert 49000 < len(m["pos"]) < 51000
    data = gpu_codec.encode_mesh(**m)
    assert data == _oracle_bytes(oracle, m)
    check_roundtrip(oracle, m, data)
    _check_decoded(oracle, data, gpu_codec.decode_mesh_batch([data])[0])


def test_gpu_lane_per_walker_at_bench_size(oracle):
    """What large batches run (lane-per-walker edgebreaker walk / attribute traversals, lane-per-stream entropy coder) on
    100k-vertex frames, forced through the environment switches: bit-identical to the default kernels and to the oracle."""
    import subprocess, sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import synth, uvol\nimport oracle as o\n"
        "o.lib(); c = uvol.Codec(device=0)\n"
        "frames = [synth.sphere_mesh(frame=k, seed=k) for k in range(5)]\n"
        "res = c.encode_mesh_batch(frames)\n"
        "f = frames[0]\n"
        "assert res[0] == o.drc_encode(f['pos'], f['idx_pos'], f.get('uv'), f.get('idx_uv'), f.get('nrm'), f.get('idx_nrm'))\n"
        "import hashlib; print('ok', hashlib.sha1(b''.join(res)).hexdigest())\n"
    ) % (os.path.join(ROOT, "universal-volumetric_amd"), os.path.join(ROOT, "oracle"))
    outs = []
    for env in (dict(), dict(UVOL_SIMT_W="16", UVOL_ENTROPY_W="8"), dict(UVOL_SIMT_W="3", UVOL_ENTROPY_WAVE="1")):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2]


def test_gpu_compact_workspace_overflow_is_retried(oracle, gpu_codec):
    """A mesh with far more corner-table vertices than input values overflows the compact per-vertex arrays on the device and
    is re-encoded alone with worst-case sizes; its batch neighbours are unaffected (see the shim test of the same name)."""
    import synth
    rng = np.random.default_rng(1)
    pos = rng.random((12, 3)).astype(np.float32)
    idx = rng.integers(0, 12, size=(6000, 3)).astype(np.uint32).reshape(-1)
    t = synth.torus_mesh()
    res = gpu_codec.encode_mesh_batch([t, dict(pos=pos, idx_pos=idx), t])
    assert res[1] == oracle.drc_encode(pos, idx, None, None, None, None)
    assert res[0] == res[2] == _oracle_bytes(oracle, t)
    assert gpu_codec.mesh_workspace(**synth.sphere_mesh()) < 80e6


def test_gpu_partitioned_dedup_duplicates_and_overflow_retry(oracle, monkeypatch):
    """Partitioned dedup on the GPU: duplicated values (lowest index wins, like the oracle), and the GEO_E_DD_OVERFLOW retry through
    the hash-table kernels when the LDS table is made too small (UVOL_DD_SLOTS=4)."""
    import synth, uvol
    from test_hipemu_geom import _dup_mesh
    m = _dup_mesh(); t = synth.torus_mesh(); s = synth.sphere_mesh(120, 61)
    want = [oracle.drc_encode(x["pos"], x["idx_pos"], x["uv"], x["idx_uv"], x["nrm"], x["idx_nrm"]) for x in (m, t, s)]
    cd = uvol.Codec(device=0)
    try:
        assert cd.encode_mesh_batch([m, t, s]) == want
        monkeypatch.setenv("UVOL_DD_SLOTS", "4")
        assert cd.encode_mesh_batch([m, t, s]) == want
    finally:
        cd.close()


def test_gpu_random_soups_and_shuffled_order_match_oracle(oracle, gpu_codec):
    """Seeded adversarial soups (non-manifold, duplicate / flipped / degenerate faces, bitwise duplicate values, random attribute
    indices, optional uv / normals) in ragged batches, and a scan-like (shuffled) storage order of a regular mesh: the partitioned
    dedup / bucket build, the per-face renumbering and the lane-per-walker kernels give the oracle's bytes or the same refusal."""
    import synth
    rng = np.random.default_rng(123)
    for batch in range(6):
        ms = []
        for k in range(5):
            seed = 100 * batch + k
            ms.append(synth.random_soup_mesh(seed, n_pos=int(rng.integers(4, 90)), n_faces=int(rng.integers(1, 260)), dup_frac=float(rng.choice([0.0, 0.3, 0.9])),
                                             with_uv=bool(rng.integers(0, 2)), with_nrm=bool(rng.integers(0, 2))))
        got = gpu_codec.encode_mesh_batch(ms, raise_on_error=False)
        for m, g in zip(ms, got):
            try:
                want = oracle.drc_encode(m["pos"], m["idx_pos"], m.get("uv"), m.get("idx_uv"), m.get("nrm"), m.get("idx_nrm"))
            except Exception:
                want = None
            assert g == want
    big = [synth.random_soup_mesh(900 + k, n_pos=3000, n_faces=9000, dup_frac=0.3) for k in range(2)]
    sh = synth.shuffle_mesh(synth.sphere_mesh(120, 61), seed=3)
    frames = big + [sh]
    got = gpu_codec.encode_mesh_batch(frames)
    for m, g in zip(frames, got):
        assert g == oracle.drc_encode(m["pos"], m["idx_pos"], m.get("uv"), m.get("idx_uv"), m.get("nrm"), m.get("idx_nrm"))


def test_gpu_storage_order_does_not_change_the_bytes(oracle):
    """a surface stored in lattice order and the same surface in a scan-like (shuffled) order both give the oracle's
    bytes at the bench's frame size - with the locality relabelling decided per frame (default: the lattice frame skips it, the
    shuffled one gets it), forced on and switched off, through the LDS walkers and the lane-per-walker kernels."""
    import subprocess, sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import synth, uvol\nimport oracle as o\n"
        "o.lib(); c = uvol.Codec(device=0)\n"
        "a = synth.sphere_mesh(frame=1, seed=1); b = synth.shuffle_mesh(a, seed=7); t = synth.shuffle_mesh(synth.torus_mesh(), seed=2)\n"
        "frames = [a, b, t, synth.shuffle_mesh(synth.grid_mesh(), seed=4)]\n"
        "res = c.encode_mesh_batch(frames)\n"
        "for f, r in zip(frames, res):\n"
        "    assert r == o.drc_encode(f['pos'], f['idx_pos'], f.get('uv'), f.get('idx_uv'), f.get('nrm'), f.get('idx_nrm'))\n"
        "print('ok')\n"
    ) % (os.path.join(ROOT, "universal-volumetric_amd"), os.path.join(ROOT, "oracle"))
    for env in (dict(), dict(UVOL_RELABEL="1"), dict(UVOL_RELABEL="0"), dict(UVOL_SIMT_W="16"), dict(UVOL_RELABEL="1", UVOL_SIMT_W="3")):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])


def test_gpu_mesh_decoder_survives_corrupt_input(oracle):
    """Bit flips, truncation and overwritten words in .drc files ON THE DEVICE, with UVOL_DEBUG=1 (every launch synchronised, a fault
    is attributed to its kernel): a decoded result or a clean error, never a fault, and the context still encodes afterwards."""
    import subprocess, sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, synth, uvol\nimport oracle as o\n"
        "o.lib(); cd = uvol.Codec(device=0)\n"
        "bases = []\n"
        "for m in (synth.torus_mesh(), synth.grid_mesh(), synth.sphere_mesh(40, 21, charts=(5, 4))):\n"
        "    bases.append((m, o.drc_encode(m['pos'], m['idx_pos'], m['uv'], m['idx_uv'], m['nrm'], m['idx_nrm'])))\n"
        "rng = np.random.default_rng(11); rej = 0; okc = 0\n"
        "for it in range(60):\n"
        "    base = bases[it %% 3][1]; b = bytearray(base); mode = (it // 3) %% 3\n"
        "    if mode == 0:\n"
        "        for _ in range(int(rng.integers(1, 4))): b[int(rng.integers(11, len(b)))] ^= 1 << int(rng.integers(0, 8))\n"
        "    elif mode == 1: b = b[:int(rng.integers(12, len(b)))]\n"
        "    else:\n"
        "        i = int(rng.integers(12, len(b) - 4)); b[i:i + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))\n"
        "    try:\n"
        "        cd.decode_mesh_batch([bytes(b), base]); okc += 1\n"
        "    except uvol.UvolError: rej += 1\n"
        "assert rej > 0\n"
        "m, base = bases[0]\n"
        "assert cd.encode_mesh(**m) == base\n"
        "print('ok', rej, okc)\n"
    ) % (os.path.join(ROOT, "universal-volumetric_amd"), os.path.join(ROOT, "oracle"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, UVOL_DEBUG="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout and "FAILED" not in r.stderr, (r.stdout[-500:], r.stderr[-2000:])


def test_gpu_256_full_size_frames_in_one_batch(oracle, gpu_codec):
    """the regime the bench runs in - hundreds of 100k-vertex frames in ONE call (lifetime-shared workspaces, the
    lane-per-walker kernels chosen by the batch size, not by an environment switch) - against the oracle's bytes for a sample of the
    frames, lattice and scan-like storage orders mixed in one batch; every other frame must equal the sampled frame of its content."""
    import synth
    base = [synth.sphere_mesh(frame=k, seed=k) for k in range(4)]
    base += [synth.shuffle_mesh(base[1], seed=9), synth.shuffle_mesh(base[2], seed=10)]
    n = 264
    frames = [base[i % len(base)] for i in range(n)]
    res = gpu_codec.encode_mesh_batch(frames)
    want = {}
    for i in (0, 1, 4, 5, 130, 263):
        k = i % len(base); f = base[k]
        want[k] = want.get(k) or oracle.drc_encode(f["pos"], f["idx_pos"], f["uv"], f["idx_uv"], f["nrm"], f["idx_nrm"])
        assert res[i] == want[k], i
    first = {}
    for i, r in enumerate(res):
        k = i % len(base)
        first.setdefault(k, r)
        assert r == first[k], (i, k)


def test_gpu_1280_distinct_full_size_frames_in_one_call(oracle):
    """the regime the bench runs in - more than 1200 frames of ~100 k vertices in ONE call, every frame with its OWN
    connectivity (vertex / face counts, chart seams and quad diagonals differ; positions differ) - byte-checked against the oracle
    on a sample spread over the call, every frame decodable in size (non-empty, plausible length)."""
    import synth, uvol
    n = 1280
    frames = synth.distinct_meshes(n, bases=16)
    assert len({(len(f["pos"]), len(f["idx_pos"])) for f in frames}) >= 12 and not np.array_equal(frames[0]["idx_pos"], frames[16]["idx_pos"])
    cd = uvol.Codec(device=0, max_batch=n)
    try:
        res = cd.encode_mesh_batch(frames)
    finally:
        cd.close()
    assert len(res) == n and all(100_000 < len(r) < 600_000 for r in res)
    for i in (0, 1, 15, 16, 17, 333, 640, 641, 1000, 1201, 1278, 1279):
        assert res[i] == _oracle_bytes(oracle, frames[i]), i


def test_gpu_resident_decode_then_encode_without_a_host_copy(oracle, gpu_codec):
    """VERDICT r3 #9 / SURVEY 8(b): uvol_decode_mesh_batch_dev leaves the decoded arrays in caller-owned HBM buffers (plain hipMalloc
    through ctypes: the caller need not be torch), uvol_encode_mesh_batch_dev_out reads them - ordered after the producer's stream, no host
    wait - and leaves the .drc bitstreams in a caller-owned HBM buffer.  The geometry never visits the host between the two calls; the
    bytes are the oracle's for the decoded arrays."""
    import ctypes as C, synth, uvol
    hip = C.CDLL("libamdhip64.so")
    def dmalloc(nbytes):
        p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), C.c_size_t(max(nbytes, 16))) == 0; return p.value
    def d2h(ptr, dtype, count):
        a = np.empty(count, dtype); assert hip.hipMemcpy(C.c_void_p(a.ctypes.data), C.c_void_p(ptr), C.c_size_t(a.nbytes), C.c_int(2)) == 0; return a
    src = [synth.sphere_mesh(120, 61, charts=(12, 6), frame=3), synth.torus_mesh(), synth.sphere_mesh(283, 177)]
    files = gpu_codec.encode_mesh_batch(src)
    n = len(files)
    metas = (uvol.DecodedMesh * n)(); owned = []
    for i, f in enumerate(files):
        nf, mv = gpu_codec.drc_info(f)
        metas[i].cap_faces = nf; metas[i].cap_values = mv
        for k, nb in (("pos", 12 * mv), ("uv", 8 * mv), ("nrm", 12 * mv), ("idx_pos", 12 * nf), ("idx_uv", 12 * nf), ("idx_nrm", 12 * nf)):
            ptr = dmalloc(nb); owned.append(ptr); setattr(metas[i], k, ptr)
    assert gpu_codec.decode_mesh_batch_dev(files, metas) == [0] * n
    meshes = (uvol.Mesh * n)()
    for i in range(n):
        m, mm = metas[i], meshes[i]
        mm.pos, mm.n_pos, mm.uv, mm.n_uv, mm.nrm, mm.n_nrm = m.pos, m.n_pos, m.uv, m.n_uv, m.nrm, m.n_nrm
        mm.idx_pos, mm.idx_uv, mm.idx_nrm, mm.n_faces = m.idx_pos, m.idx_uv, m.idx_nrm, m.n_faces
    cap = 8 << 20; out = dmalloc(cap); owned.append(out)
    stream = C.c_void_p(); assert hip.hipStreamCreate(C.byref(stream)) == 0
    assert hip.hipMemsetAsync(C.c_void_p(out), C.c_int(0), C.c_size_t(cap), stream) == 0       # work queued on the producer's stream: the encode is ordered behind it
    offs, lens, st = gpu_codec.encode_mesh_batch_dev_out(meshes, out, cap, producer_stream=stream.value)
    assert st == [0] * n
    got = d2h(out, np.uint8, cap)
    for i in range(n):
        m = metas[i]
        want = oracle.drc_encode(d2h(m.pos, np.float32, 3 * m.n_pos).reshape(-1, 3), d2h(m.idx_pos, np.uint32, 3 * m.n_faces), d2h(m.uv, np.float32, 2 * m.n_uv).reshape(-1, 2), d2h(m.idx_uv, np.uint32, 3 * m.n_faces),
                                 d2h(m.nrm, np.float32, 3 * m.n_nrm).reshape(-1, 3), d2h(m.idx_nrm, np.uint32, 3 * m.n_faces))
        assert got[offs[i]:offs[i] + lens[i]].tobytes() == want, i
    hip.hipStreamDestroy(stream)
    for ptr in owned:
        hip.hipFree(C.c_void_p(ptr))


def test_gpu_obj_text_parsed_on_the_device(oracle, gpu_codec, tmp_path):
    """SURVEY 8 f-3 / VERDICT r3 #8 on the GPU: the grammar cases of the emulation test, and three 100 k-vertex files of 20 MB of text
    each (the shape tools/e2e_files.py writes) parsed in one call - arrays bit-identical to the host parser's (itself pinned to strtof),
    encoded straight from HBM to the oracle's bytes."""
    import synth
    from test_hipemu_geom import _obj_texts, _host_obj, _dev_array
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    def d2h(ptr, dtype, count):
        a = np.empty(count, dtype)
        if count: assert hip.hipMemcpy(C.c_void_p(a.ctypes.data), C.c_void_p(ptr), C.c_size_t(a.nbytes), C.c_int(2)) == 0
        return a
    paths = _obj_texts(tmp_path)
    for k in range(3):
        m = synth.sphere_mesh(frame=k, seed=k)
        ip, iu, inn = (m[key].reshape(-1, 3).astype(np.int64) + 1 for key in ("idx_pos", "idx_uv", "idx_nrm"))
        p = tmp_path / ("big%d.obj" % k)
        with open(p, "w") as f:
            f.write("\n".join("v %.6f %.6f %.6f" % tuple(v) for v in m["pos"].tolist())); f.write("\n")
            f.write("\n".join("vt %.7f %.7f" % tuple(v) for v in m["uv"].tolist())); f.write("\n")
            f.write("\n".join("vn %.6f %.6f %.6f" % tuple(v) for v in m["nrm"].tolist())); f.write("\n")
            f.write("\n".join("f %d/%d/%d %d/%d/%d %d/%d/%d" % tuple(r) for r in np.stack([ip, iu, inn], -1).reshape(-1, 9).tolist())); f.write("\n")
        paths.append(p)
    texts = [open(p, "rb").read() for p in paths]
    meshes, st = gpu_codec.parse_obj_batch_dev(texts, slot=0)
    assert st == [0] * len(texts), st
    got = gpu_codec.encode_mesh_batch_dev(meshes)
    for p, m, g in zip(paths, meshes, got):
        h = _host_obj(p)
        assert (m.n_pos, m.n_faces) == (len(h["pos"]), len(h["idx_pos"]) // 3), p
        assert np.array_equal(d2h(m.pos, np.uint32, 3 * m.n_pos), h["pos"].reshape(-1).view(np.uint32)) and np.array_equal(d2h(m.idx_pos, np.uint32, 3 * m.n_faces), h["idx_pos"]), p
        if len(h["uv"]):
            assert np.array_equal(d2h(m.uv, np.uint32, 2 * m.n_uv), h["uv"].reshape(-1).view(np.uint32)) and np.array_equal(d2h(m.idx_uv, np.uint32, 3 * m.n_faces), h["idx_uv"]), p
        if len(h["nrm"]):
            assert np.array_equal(d2h(m.nrm, np.uint32, 3 * m.n_nrm), h["nrm"].reshape(-1).view(np.uint32)) and np.array_equal(d2h(m.idx_nrm, np.uint32, 3 * m.n_faces), h["idx_nrm"]), p
        if p.name != "d.obj":                   # (d.obj: random numbers as positions, a degenerate soup - parsed, not worth an oracle run)
            assert g == oracle.drc_encode(h["pos"], h["idx_pos"], h["uv"] if len(h["uv"]) else None, h["idx_uv"] if len(h["uv"]) else None, h["nrm"] if len(h["nrm"]) else None, h["idx_nrm"] if len(h["nrm"]) else None), p


def test_gpu_batch_sizes_alternate_on_one_context(oracle):
    """Batches above 1200 frames join the auxiliary stream (valence replay) before the attribute record tables are written, its
    inputs sharing their bytes; smaller batches join it before the entropy stage and give those arrays longer lifetimes.  The
    workspace plan is cached per frame shape - and must be keyed by that choice too: a small batch after a large one of the same
    shape once inherited the large batch's plan and raced (found by the bench's variants).  Same shape, 1300 frames then 7, then
    1300 again, one context: bytes of the oracle every time."""
    import synth, uvol
    frames = [synth.sphere_mesh(40, 21, charts=(5, 4), frame=k % 4) for k in range(1300)]
    want = [oracle.drc_encode(f["pos"], f["idx_pos"], f.get("uv"), f.get("idx_uv"), f.get("nrm"), f.get("idx_nrm")) for f in frames[:4]]
    cd = uvol.Codec(device=0, max_batch=1300)
    try:
        for n in (1300, 7, 1300, 150):
            got = cd.encode_mesh_batch(frames[:n])
            assert all(got[i] == want[i % 4] for i in range(n)), n
        cd.trim()                                             # uvol_trim: the workspaces go back to the device, the context goes on
        got = cd.encode_mesh_batch(frames[:300])
        assert all(got[i] == want[i % 4] for i in range(300))
    finally:
        cd.close()


def test_gpu_enqueue_form_of_the_abi(oracle, gpu_codec):
    """uvol_encode_mesh_batch_async / uvol_encode_texture_segments_async + uvol_sync on the device: several calls enqueued back to
    back, results equal to the blocking entry points' (and the oracle's), a failing frame fails alone."""
    import synth
    a, b = synth.sphere_mesh(120, 61, charts=(12, 6), frame=3), synth.grid_mesh()
    bad = dict(b, idx_pos=b["idx_pos"].copy()); bad["idx_pos"][5] = 10 ** 6
    tex = synth.texture_sequence(2, size=64, seed=1)
    gpu_codec.start_mesh_batch([a, b]); gpu_codec.start_mesh_batch([b, bad, a]); gpu_codec.start_texture_segments([tex, tex])
    r = gpu_codec.finish()
    ea, eb = _oracle_bytes(oracle, a), _oracle_bytes(oracle, b)
    assert r[0] == [ea, eb] and r[1] == [eb, None, ea] and r[2] == [oracle.ktx2_encode(tex)] * 2


def test_gpu_enqueued_device_calls_round_the_lane_ring(oracle):
    """Round 5: an enqueued call on DEVICE inputs is cut into four groups while the context's ring has six lanes, so the groups of the next
    call start on the lanes the previous one left free and the ring wraps inside a stream of calls (the lanes past the first call's get
    their buffers ahead of time).  Four calls of 700 small distinct frames resident in HBM, alternating output slots: every call's bytes
    are the oracle's, frame by frame."""
    import torch, synth, uvol
    frames = synth.distinct_meshes(7, 24, 13, bases=3, charts=(3, 2))
    want = [_oracle_bytes(oracle, f) for f in frames]
    n = 700
    keep, meshes = [], []
    for f in frames:
        m, arrs = uvol.Codec._mesh_host(**f)
        dev = [None if a is None else torch.from_numpy(a).cuda() for a in arrs]
        keep.append(dev)
        pos, uv, nrm, ip, iu, inr = dev
        m.pos = pos.data_ptr(); m.idx_pos = ip.data_ptr()
        if uv is not None: m.uv = uv.data_ptr(); m.idx_uv = iu.data_ptr()
        if nrm is not None: m.nrm = nrm.data_ptr(); m.idx_nrm = inr.data_ptr()
        meshes.append(m)
    torch.cuda.synchronize()
    arr = (uvol.Mesh * n)(*[meshes[i % 7] for i in range(n)])
    cd = uvol.Codec(device=0, max_batch=n)
    try:
        for k in range(4):
            cd.start_mesh_batch_dev(arr, slot=k & 1)
        res = cd.finish()
    finally:
        cd.close()
    assert len(res) == 4
    for rr in res[2:]:                                      # (the views of calls 0 / 1 were overwritten by calls 2 / 3 of the same slots: same content)
        assert len(rr) == n
        for i, r in enumerate(rr):
            assert r is not None and bytes(r) == want[i % 7], i


def test_gpu_decoder_reads_the_other_draco_tool_sets(oracle, gpu_codec):
    """streams with the STANDARD edgebreaker traversal (stock compression levels 1..5) and with SEQUENTIAL connectivity
    (level 0) decode on the device to exactly what the CPU restatement decodes - at test sizes and at 100k vertices -,
    mixed with valence-edgebreaker files in one batch."""
    import synth
    ms = [synth.torus_mesh(), synth.grid_mesh(), synth.sphere_mesh(frame=2, seed=2)]
    files = []
    for f in ms:
        for method in (0, 1, 2, 3):                     # 3: sequential connectivity with compressed indices (connectivity_method 0)
            files.append(oracle.drc_encode(f["pos"], f["idx_pos"], f["uv"], f["idx_uv"], f["nrm"], f["idx_nrm"], method=method))
    t = ms[0]
    files.append(oracle.drc_encode(t["pos"], t["idx_pos"], method=2))
    files.append(oracle.drc_encode(t["pos"], t["idx_pos"], method=3))
    from test_hipemu_geom import _check_decoded
    for data, got in zip(files, gpu_codec.decode_mesh_batch(files)):
        _check_decoded(oracle, data, got)


def test_gpu_sequential_connectivity_at_compression_level_0(oracle):
    """SURVEY row a3b on the device: DRACO_COMPRESSION_LEVEL 0 = sequential connectivity.  Bytes equal the CPU restatement's (method 2)
    at test sizes and at 100k vertices (lattice and shuffled storage), and the frames round-trip (same triangles, positions within
    half a quantisation step) through the fixture-pinned decoder's sequential branch."""
    import synth, uvol
    cd = uvol.Codec(device=0, DRACO_COMPRESSION_LEVEL=0)
    try:
        big = synth.sphere_mesh(frame=1, seed=1)
        ms = [synth.torus_mesh(), synth.grid_mesh(), big, synth.shuffle_mesh(big, seed=5), dict(pos=big["pos"], idx_pos=big["idx_pos"])]
        got = cd.encode_mesh_batch(ms)
        for m, g in zip(ms, got):
            assert g == oracle.drc_encode(m["pos"], m["idx_pos"], m.get("uv"), m.get("idx_uv"), m.get("nrm"), m.get("idx_nrm"), method=2)
        check_roundtrip(oracle, ms[2], got[2]); check_roundtrip(oracle, ms[0], got[0])
        from test_hipemu_geom import _check_decoded
        for data, dec in zip(got, cd.decode_mesh_batch(got)):
            _check_decoded(oracle, data, dec)
    finally:
        cd.close()
