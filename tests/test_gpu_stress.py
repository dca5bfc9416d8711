"""Alternating workloads on ONE context (GPU): what the fixed-shape tests cannot see - state carried from one call to the next (cached
workspace plans, grown buffers, stream joins chosen by batch size, the alpha retry of the texture path).  Seeded, a few seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _want_mesh(oracle, f, **kw):
    return oracle.drc_encode(f["pos"], f["idx_pos"], f.get("uv"), f.get("idx_uv"), f.get("nrm"), f.get("idx_nrm"), **kw)


def test_gpu_geometry_calls_of_changing_size_and_shape_on_one_context(oracle):
    """Batch sizes on both sides of every threshold the host decides by (1 frame, the wave-per-walker / lane-per-walker switch, the
    late / early join at 1200, the coherence look-ahead at 256), shapes that repeat (plan cache hits) and that change, shuffled storage
    order in between, blocking and enqueue form mixed, then the decoder on the same context: bytes of the oracle every time."""
    import synth, uvol
    rng = np.random.default_rng(20260929)
    shapes = [synth.sphere_mesh(40, 21, charts=(5, 4), frame=k) for k in range(3)] + [synth.torus_mesh(16, 8), synth.grid_mesh(), synth.sphere_mesh(24, 13, charts=(3, 2), crease=False)]
    shapes.append(synth.shuffle_mesh(shapes[0], seed=5))
    want = [_want_mesh(oracle, f) for f in shapes]
    cd = uvol.Codec(device=0, max_batch=1400)
    try:
        for n in (1, 1300, 7, 255, 257, 1201, 1200, 3, 600, 1, 1300, 64):
            pick = rng.integers(0, len(shapes), size=n) if n % 2 else np.full(n, rng.integers(0, len(shapes)))
            frames = [shapes[i] for i in pick]
            if n in (7, 600):
                cd.start_mesh_batch(frames); got = cd.finish()[0]
            else:
                got = cd.encode_mesh_batch(frames)
            bad = [i for i in range(n) if got[i] != want[pick[i]]]
            assert not bad, (n, bad[:5])
            if n in (3, 64):
                dec = cd.decode_mesh_batch(got)
                from test_hipemu_geom import _check_decoded
                for data, d in zip(got[:8], dec[:8]):
                    _check_decoded(oracle, data, d)
    finally:
        cd.close()


def test_gpu_texture_calls_of_changing_size_and_kind_on_one_context(oracle):
    """Segment counts 1 ... 40, layer counts 1 ... 5, two image sizes, opaque and alpha segments mixed in a batch (the alpha segments are
    encoded again inside the call), UASTC contexts beside it, the decoder in between: bytes of the oracle every time."""
    import zlib
    import synth, uvol
    from test_hipemu_tex import _alpha_sequence
    rng = np.random.default_rng(7)
    cd = uvol.Codec(device=0)
    try:
        for size, nl, nseg in ((64, 3, 1), (64, 3, 40), (96, 5, 6), (64, 1, 9), (96, 2, 1), (64, 3, 17)):
            segs = []
            for s in range(nseg):
                alpha = rng.random() < 0.4
                t = _alpha_sequence(nl, size, int(rng.integers(1, 50))) if alpha else synth.texture_sequence(nl, size=size, seed=int(rng.integers(1, 50)))
                segs.append(t)
            got = cd.encode_texture_segments(segs)
            distinct = {}
            for s, (t, g) in enumerate(zip(segs, got)):
                key = tuple(zlib.crc32(a.tobytes()) for a in t)
                if key not in distinct:
                    distinct[key] = oracle.ktx2_encode(t)
                assert g == distinct[key], (size, nl, nseg, s)
            d = oracle.ktx2_decode(got[0])
            dec = cd.decode_texture_segments(got[:1])[0]
            for l in range(nl):
                assert np.array_equal(dec[l], d.images[l])
    finally:
        cd.close()


def test_gpu_uplink_pinned_inputs_enqueued_calls(oracle):
    """Round 6: inputs in uvol_host_alloc memory travel through the context's uplink - slots filled on a copy stream
    when the call begins, lanes ordered behind them by events, slots re-used behind release events.  On the real streams: geometry calls of
    700 frames (4 groups) enqueued three deep on a ring of six slots, frames laid out back to back in the arena (one DMA per run), then a
    call with one pageable array (staged path) on the same context; texture calls of 130 segments (two parts) enqueued three deep with an
    alpha segment in the deferred last part.  Every byte is the oracle's."""
    import synth, uvol
    from test_hipemu_tex import _alpha_sequence
    shapes = [synth.sphere_mesh(40, 21, charts=(5, 4), frame=k) for k in range(3)] + [synth.torus_mesh(16, 8), synth.grid_mesh()]
    want = [_want_mesh(oracle, f) for f in shapes]
    ar = uvol.PinnedArena(256 << 20)
    cd = uvol.Codec(device=0, max_batch=800)
    ct = uvol.Codec(device=0)
    try:
        pm = [{k: ar.put(v) for k, v in f.items()} for f in shapes]
        n = 700
        frames = [pm[i % len(pm)] for i in range(n)]
        for _ in range(3):
            cd.start_mesh_batch(frames)
        res = cd.finish()
        assert len(res) == 3
        for got in res:
            bad = [i for i in range(n) if got[i] != want[i % len(pm)]]
            assert not bad, bad[:5]
        got = cd.encode_mesh_batch(frames[:330])                                     # blocking: two groups
        assert all(got[i] == want[i % len(pm)] for i in range(330))
        mixed = [dict(frames[0], pos=np.array(shapes[0]["pos"]))] + frames[1:200]      # one pageable array: the whole call is staged
        got = cd.encode_mesh_batch(mixed)
        assert all(got[i] == want[i % len(pm)] for i in range(200))
        tex = [synth.texture_sequence(2, size=64, seed=k) for k in range(3)] + [_alpha_sequence(2, 64, 7)]
        want_t = [oracle.ktx2_encode(t) for t in tex]
        pt = [[ar.put(a) for a in t] for t in tex]
        ns = 130
        segs = [pt[s % 3] for s in range(ns - 1)] + [pt[3]]                           # the alpha segment closes the call's last part
        for _ in range(3):
            ct.start_texture_segments(segs)
        res = ct.finish()
        assert len(res) == 3
        for got in res:
            assert all(got[s] == want_t[s % 3] for s in range(ns - 1)) and got[ns - 1] == want_t[3]
        got = ct.encode_texture_segments(segs[:5])
        assert all(got[s] == want_t[s % 3] for s in range(5))
    finally:
        cd.close(); ct.close(); ar.close()
