"""synth.py — seeded synthetic UVOL frames: OBJ-shaped meshes and RGBA8 textures.

Geometry: displaced UV-sphere "body" in millimetres (bbox ~0.6 x 1.8 x 0.5 m like the reference
fixture, example/src/VolumetricPlayer.tsx:180 scales by 0.001), closed manifold, UV atlas cut into
rectangular charts (so UV seams exist), per-vertex normals plus one hard crease ring (so the normal
attribute has a few seams, like the fixtures' 25 seam corners).
Texture: multi-octave value noise over ~70 % of the image, flat black background, frame-to-frame
change confined to ~30 % of the 4x4 blocks.
"""
import numpy as np


def sphere_mesh(n_seg=400, n_ring=251, frame=0, charts=(40, 25), seed=0, crease=True):
    """Closed UV-sphere: V = n_seg*(n_ring-1)+2, F = 2*n_seg*(n_ring-1).
    Defaults give V=100,002 / F=200,000 (BASELINE config); (283,177) -> ~50k verts."""
    rng = np.random.default_rng(seed)
    nr = n_ring - 1                                   # interior rings
    th = (np.arange(n_seg) / n_seg) * 2 * np.pi       # longitude
    ph = (np.arange(1, n_ring) / n_ring) * np.pi      # colatitude, poles excluded
    T, Pp = np.meshgrid(th, ph)                       # (nr, n_seg)
    t = frame * 0.07
    rad = 1.0 + 0.08 * np.sin(3 * T + t) * np.sin(2 * Pp) + 0.05 * np.cos(5 * Pp - 2 * t) + 0.02 * np.sin(9 * T + 7 * Pp + 3 * t)
    x = 300.0 * rad * np.sin(Pp) * np.cos(T)
    y = 900.0 + 880.0 * rad * np.cos(Pp)
    z = -300.0 + 250.0 * rad * np.sin(Pp) * np.sin(T)
    body = np.stack([x, y, z], -1).reshape(-1, 3)
    jitter = rng.standard_normal(body.shape).astype(np.float32) * 0.15
    top = np.array([[0.0, 900.0 + 880.0 * (1.0 + 0.05 * np.cos(-2 * t)), -300.0]])
    bot = np.array([[0.0, 900.0 - 880.0 * (1.0 + 0.05 * np.cos(5 * np.pi - 2 * t)), -300.0]])
    pos = np.concatenate([body + jitter, top, bot]).astype(np.float32)
    i_top, i_bot = nr * n_seg, nr * n_seg + 1

    def vid(r, s):
        return r * n_seg + (s % n_seg)

    faces = []
    r = np.arange(nr - 1)[:, None]; s = np.arange(n_seg)[None, :]
    a = vid(r, s); b = vid(r, s + 1); c = vid(r + 1, s); d = vid(r + 1, s + 1)
    quads1 = np.stack([a, c, b], -1).reshape(-1, 3)
    quads2 = np.stack([b, c, d], -1).reshape(-1, 3)
    body_f = np.empty((quads1.shape[0] * 2, 3), dtype=np.int64)
    body_f[0::2] = quads1; body_f[1::2] = quads2
    s1 = np.arange(n_seg)
    cap_top = np.stack([np.full(n_seg, i_top), vid(0, s1), vid(0, s1 + 1)], -1)
    cap_bot = np.stack([np.full(n_seg, i_bot), vid(nr - 1, s1 + 1), vid(nr - 1, s1)], -1)
    idx_pos = np.concatenate([cap_top, body_f, cap_bot]).astype(np.uint32)
    nf = len(idx_pos)

    # ---- UV atlas: charts over (segment, ring) space; every face gets the chart of its low corner ----
    cs, cr = charts
    seg_per = max(1, n_seg // cs); ring_per = max(1, nr // cr)
    # face -> (r0, s0) lattice cell
    fr = np.concatenate([np.zeros(n_seg, int), np.repeat(np.arange(nr - 1), 2 * n_seg), np.full(n_seg, nr - 1)])
    fs = np.concatenate([s1, np.repeat(np.tile(s1, nr - 1), 2).reshape(nr - 1, n_seg, 2).reshape(-1) if False else np.tile(np.repeat(s1, 2), nr - 1), s1])
    chart_s = np.minimum(fs // seg_per, cs - 1); chart_r = np.minimum(fr // ring_per, cr - 1)
    chart = chart_r * cs + chart_s
    # per-corner lattice coordinates (unwrapped in s so a chart never straddles the 2*pi seam)
    pr = np.where(idx_pos >= nr * n_seg, -1, idx_pos // n_seg).astype(np.int64)
    ps = (idx_pos % n_seg).astype(np.int64)
    ps = np.where((ps < fs[:, None]) & (idx_pos < nr * n_seg), ps + n_seg, ps)        # wrap-around column
    pole_top = idx_pos == i_top; pole_bot = idx_pos == i_bot
    pr = np.where(pole_top, -1, np.where(pole_bot, nr, pr)); ps = np.where(pole_top | pole_bot, fs[:, None], ps)
    cell_w, cell_h = 1.0 / cs, 1.0 / cr
    u = (chart_s[:, None] + 0.04 + 0.92 * (ps - chart_s[:, None] * seg_per) / (seg_per + (n_seg - cs * seg_per) + 1)) * cell_w
    v = (chart_r[:, None] + 0.04 + 0.92 * (pr + 1 - chart_r[:, None] * ring_per) / (ring_per + (nr - cr * ring_per) + 2)) * cell_h
    key = (chart[:, None].astype(np.int64) << 40) | ((pr + 1).astype(np.int64) << 20) | ps.astype(np.int64)
    uniq, inv = np.unique(key.reshape(-1), return_inverse=True)
    uv = np.zeros((len(uniq), 2), dtype=np.float32)
    uv[inv, 0] = u.reshape(-1); uv[inv, 1] = v.reshape(-1)
    idx_uv = inv.reshape(nf, 3).astype(np.uint32)

    # ---- normals: area-weighted per vertex + noise; optional hard crease along the middle ring ----
    P = pos.astype(np.float64)
    fn = np.cross(P[idx_pos[:, 1]] - P[idx_pos[:, 0]], P[idx_pos[:, 2]] - P[idx_pos[:, 0]])
    vn = np.zeros_like(P)
    for k in range(3):
        np.add.at(vn, idx_pos[:, k], fn)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-12)
    vn += rng.standard_normal(vn.shape) * 0.02
    vn /= np.linalg.norm(vn, axis=1, keepdims=True)
    nrm = vn.astype(np.float32)
    idx_nrm = idx_pos.copy()
    if crease:
        rc = nr // 2                                  # faces below ring rc use a second normal set on that ring
        ring_ids = vid(rc, s1)
        extra = nrm[ring_ids] * np.float32(0.7) + np.array([0, -0.7, 0], dtype=np.float32)
        extra /= np.linalg.norm(extra, axis=1, keepdims=True)
        base = len(nrm); nrm = np.concatenate([nrm, extra.astype(np.float32)])
        remap = np.full(len(pos), -1, dtype=np.int64); remap[ring_ids] = base + np.arange(n_seg)
        below = fr >= rc
        sel = below[:, None] & (remap[idx_pos] >= 0)
        idx_nrm = np.where(sel, remap[idx_pos], idx_pos).astype(np.uint32)
    return dict(pos=pos, idx_pos=idx_pos.reshape(-1), uv=uv, idx_uv=idx_uv.reshape(-1), nrm=nrm, idx_nrm=idx_nrm.reshape(-1).astype(np.uint32))


def flip_diagonals(m, n_seg, n_ring, seed, frac=0.5, jitter=0.0):
    """A sphere_mesh() with a seeded subset of its body quads re-triangulated along the other diagonal: (a, c, b) + (b, c, d) becomes
    (a, c, d) + (a, d, b) in all three index arrays.  Same surface, same value arrays, still a closed manifold with the same seams and
    the same winding - but its own connectivity: the edgebreaker walk and the attribute traversals of two such frames take different
    paths (a capture's frames never share an index array; the reference's 250 frames have 26,144 - 27,979 vertices each).
    jitter > 0 also moves the positions by a seeded offset, so frames made from one base differ in content too."""
    rng = np.random.default_rng(seed)
    nq = n_seg * (n_ring - 2)                         # body quads, stored as face pairs right behind the top cap
    pick = np.flatnonzero(rng.random(nq) < frac)
    out = dict(m)
    for key in ("idx_pos", "idx_uv", "idx_nrm"):
        if m.get(key) is None:
            continue
        f = np.array(m[key]).reshape(-1, 3)
        t1 = f[n_seg + 2 * pick]; t2 = f[n_seg + 2 * pick + 1]
        assert np.array_equal(t1[:, 2], t2[:, 0]) and np.array_equal(t1[:, 1], t2[:, 1]), "quad halves do not share their diagonal"
        a, c, b, d = t1[:, 0], t1[:, 1], t1[:, 2], t2[:, 2]
        f[n_seg + 2 * pick] = np.stack([a, c, d], -1); f[n_seg + 2 * pick + 1] = np.stack([a, d, b], -1)
        out[key] = f.reshape(-1)
    if jitter > 0:
        out["pos"] = (np.asarray(m["pos"], np.float32) + rng.standard_normal(np.asarray(m["pos"]).shape).astype(np.float32) * np.float32(jitter)).astype(np.float32)
    return out


def distinct_meshes(count, n_seg=400, n_ring=251, bases=None, charts=(40, 25)):
    """`count` frames of about n_seg * (n_ring - 1) vertices, each with its OWN connectivity (SURVEY 8d's sequence made honest about
    what a capture looks like).  Tessellations differ three ways: segment / ring counts within +-2 (vertex and face counts differ from
    frame to frame), the chart grid of the UV atlas (+-1 column / row: different seams), and a seeded half of the quads flipped.
    `bases` bounds the number of sphere_mesh() calls (0.5 s each at 100 k vertices): frame k is base k % bases with its own flips and,
    from the second use of a base on, its own position jitter.  Default: one base per frame up to 64."""
    bases = min(count, 64) if bases is None else max(1, min(bases, count))
    var = [(0, 0), (1, -1), (-1, 1), (2, 0), (0, 2), (-2, 1), (1, -2), (-1, -1), (2, 2), (-2, -2), (0, -1), (-1, 0), (1, 1), (2, -1), (-2, 2), (0, 1)]
    small = n_seg < 24 or n_ring < 16                  # tiny test meshes keep their lattice (the +-2 would change them by a large factor)
    B = []
    for b in range(bases):
        ds, dr = (0, 0) if small else var[b % len(var)]
        ns, nr = n_seg + ds, n_ring + dr
        ch = (max(1, charts[0] + (b // len(var)) % 3 - 1), max(1, charts[1] + (b // (3 * len(var))) % 3 - 1))
        B.append((ns, nr, sphere_mesh(ns, nr, frame=b, seed=b, charts=ch)))
    out = []
    for k in range(count):
        ns, nr, m = B[k % bases]
        out.append(flip_diagonals(m, ns, nr, seed=7919 * k + 13, frac=0.5, jitter=0.0 if k < bases else 0.05))
    return out


def grid_mesh(nx=24, ny=16, seed=1, holes=True):
    """Open height-field patch with a boundary (and optionally a punched hole): exercises L/R/E starts and boundary fans."""
    rng = np.random.default_rng(seed)
    xs, ys = np.meshgrid(np.arange(nx), np.arange(ny))
    pos = np.stack([xs * 10.0, ys * 10.0, 15.0 * np.sin(xs * 0.4) * np.cos(ys * 0.3)], -1).reshape(-1, 3).astype(np.float32)
    pos += rng.standard_normal(pos.shape).astype(np.float32) * 0.3
    f = []
    for j in range(ny - 1):
        for i in range(nx - 1):
            if holes and 5 <= i < 9 and 4 <= j < 8:
                continue
            a, b, c, d = j * nx + i, j * nx + i + 1, (j + 1) * nx + i, (j + 1) * nx + i + 1
            f += [(a, b, c), (b, d, c)]
    idx = np.array(f, dtype=np.uint32)
    uv = (pos[:, :2] / np.array([nx * 10.0, ny * 10.0], dtype=np.float32)).astype(np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (len(pos), 1))
    nrm[:, 0] = -0.3 * np.cos(xs.reshape(-1) * 0.4); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return dict(pos=pos, idx_pos=idx.reshape(-1), uv=uv, idx_uv=idx.reshape(-1).copy(), nrm=nrm.astype(np.float32), idx_nrm=idx.reshape(-1).copy())


def torus_mesh(n_major=48, n_minor=20, seed=2):
    """Genus-1 closed mesh: forces topology-split events (handles) in the edgebreaker traversal."""
    rng = np.random.default_rng(seed)
    a = np.arange(n_major) / n_major * 2 * np.pi; b = np.arange(n_minor) / n_minor * 2 * np.pi
    A, B = np.meshgrid(a, b, indexing="ij")
    pos = np.stack([(100 + 35 * np.cos(B)) * np.cos(A), (100 + 35 * np.cos(B)) * np.sin(A), 35 * np.sin(B)], -1).reshape(-1, 3).astype(np.float32)
    pos += rng.standard_normal(pos.shape).astype(np.float32) * 0.2
    f = []
    for i in range(n_major):
        for j in range(n_minor):
            p00 = i * n_minor + j; p01 = i * n_minor + (j + 1) % n_minor
            p10 = ((i + 1) % n_major) * n_minor + j; p11 = ((i + 1) % n_major) * n_minor + (j + 1) % n_minor
            f += [(p00, p10, p01), (p01, p10, p11)]
    idx = np.array(f, dtype=np.uint32)
    # uv per corner with wrap seams
    fi = np.repeat(np.arange(n_major), n_minor * 2); fj = np.tile(np.repeat(np.arange(n_minor), 2), n_major)
    ci = (idx // n_minor).astype(np.int64); cj = (idx % n_minor).astype(np.int64)
    ci = np.where(ci < fi[:, None], ci + n_major, ci); cj = np.where(cj < fj[:, None], cj + n_minor, cj)
    key = ci * 4096 + cj
    uniq, inv = np.unique(key.reshape(-1), return_inverse=True)
    uv = np.stack([(uniq // 4096) / (n_major + 1.0), (uniq % 4096) / (n_minor + 1.0)], -1).astype(np.float32)
    P = pos.astype(np.float64)
    fn = np.cross(P[idx[:, 1]] - P[idx[:, 0]], P[idx[:, 2]] - P[idx[:, 0]])
    vn = np.zeros_like(P)
    for k in range(3):
        np.add.at(vn, idx[:, k], fn)
    vn /= np.linalg.norm(vn, axis=1, keepdims=True)
    return dict(pos=pos, idx_pos=idx.reshape(-1), uv=uv, idx_uv=inv.astype(np.uint32), nrm=vn.astype(np.float32), idx_nrm=idx.reshape(-1).copy())


def _value_noise(h, w, cells, rng):
    g = rng.random((cells + 2, cells + 2)).astype(np.float32)
    ys = np.linspace(0, cells, h, endpoint=False); xs = np.linspace(0, cells, w, endpoint=False)
    y0 = ys.astype(int); x0 = xs.astype(int); fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    fy = fy * fy * (3 - 2 * fy); fx = fx * fx * (3 - 2 * fx)
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx


def texture_sequence(n_frames, size=2048, seed=0, change_frac=0.30, bg_frac=0.30):
    """List of HxWx4 uint8 frames (alpha 255). ~bg_frac black background, ~change_frac of blocks change per frame."""
    rng = np.random.default_rng(seed)
    h = w = size
    base = np.zeros((h, w, 3), dtype=np.float32)
    for ch in range(3):
        acc = np.zeros((h, w), dtype=np.float32)
        for o, amp in ((4, 0.5), (16, 0.25), (64, 0.15), (256, 0.10)):
            acc += amp * _value_noise(h, w, min(o, size // 4), rng)
        base[..., ch] = acc
    base = (base - base.min()) / (base.max() - base.min())
    base[..., 0] = 0.25 + 0.7 * base[..., 0]; base[..., 1] = 0.15 + 0.6 * base[..., 1]; base[..., 2] = 0.1 + 0.5 * base[..., 2]
    yy, xx = np.mgrid[0:h, 0:w]
    fg = ((xx / w - 0.5) ** 2 / 0.21 + (yy / h - 0.5) ** 2 / 0.235) < 1.0        # ellipse ~70 % of the area
    if bg_frac <= 0:
        fg[:] = True
    mov = _value_noise(h // 4, w // 4, 12, rng)
    thr = np.quantile(mov, 1.0 - change_frac)
    moving_blocks = np.kron(mov > thr, np.ones((4, 4), dtype=bool))[:h, :w]
    frames = []
    for f in range(n_frames):
        img = base.copy()
        if f > 0:
            wob = 0.06 * np.sin(0.9 * f + xx / 37.0) * np.cos(0.7 * f + yy / 29.0)
            img += (wob * moving_blocks)[..., None]
        img = np.clip(img, 0, 1) * fg[..., None]
        rgba = np.empty((h, w, 4), dtype=np.uint8)
        rgba[..., :3] = (img * 255.0 + 0.5).astype(np.uint8); rgba[..., 3] = 255
        frames.append(rgba)
    return frames


def edge_case_meshes():
    """Small meshes that exercise the non-manifold / boundary / multi-component rules of the corner table
    (extra faces on an edge become boundaries, one fan per vertex instance, inconsistent orientation, duplicate faces)."""
    def mk(pos, faces):
        return dict(pos=np.array(pos, np.float32), idx_pos=np.array(faces, np.uint32).reshape(-1))
    c = {}
    c["fin"] = mk([[0, 0, 0], [1, 0, 0], [0.5, 1, 0], [0.5, -1, 0], [0.5, 0, 1]], [[0, 1, 2], [1, 0, 3], [0, 1, 4]])
    c["bowtie"] = mk([[0, 0, 0], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0]], [[0, 1, 2], [0, 3, 4]])
    c["flipped"] = mk([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], [[0, 1, 2], [1, 2, 3]])
    c["dupface"] = mk([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], [[0, 1, 2], [0, 1, 2], [2, 1, 3]])
    t = torus_mesh(10, 6); s = sphere_mesh(12, 7, charts=(2, 2))
    c["two_components"] = dict(pos=np.concatenate([t["pos"], s["pos"] + 500]), idx_pos=np.concatenate([t["idx_pos"], s["idx_pos"] + len(t["pos"])]))
    c["small_torus_handles"] = torus_mesh(16, 8)
    g = grid_mesh(7, 5, holes=False)
    c["open_grid"] = g
    return c


def shuffle_mesh(m, seed=0):
    """The same surface with scan-like storage order: a seeded permutation of the faces and of every value array (indices
    relabelled), as a photogrammetry / marching-cubes export has it — consecutive faces are no longer neighbours, so the serial
    walkers find none of a face's neighbours on the cache line they just fetched (the UV-sphere lattice stores four consecutive
    faces per 128 bytes: best case)."""
    rng = np.random.default_rng(seed)
    nf = len(m["idx_pos"]) // 3
    fperm = rng.permutation(nf)
    out = {}
    for val, idx, cols in (("pos", "idx_pos", 3), ("uv", "idx_uv", 2), ("nrm", "idx_nrm", 3)):
        if m.get(val) is None:
            continue
        a = np.asarray(m[val]).reshape(-1, cols); n = len(a)
        vperm = rng.permutation(n)                     # new position k holds old value vperm[k]
        inv = np.empty(n, np.int64); inv[vperm] = np.arange(n)
        out[val] = np.ascontiguousarray(a[vperm])
        out[idx] = np.ascontiguousarray(inv[np.asarray(m[idx]).reshape(nf, 3)[fperm]].astype(np.uint32).reshape(-1))
    return out


def random_soup_mesh(seed, n_pos=40, n_faces=120, dup_frac=0.3, with_uv=True, with_nrm=True):
    """Seeded adversarial triangle soup: faces drawn at random over few positions (non-manifold edges and vertices, duplicate and
    flipped faces, isolated components), a fraction of the value arrays duplicated bit for bit under other indices, degenerate
    faces (repeated index, or two indices whose values are equal), unused values, random per-corner uv / normal indices."""
    rng = np.random.default_rng(seed)
    def values(n, d):
        v = rng.random((n, d)).astype(np.float32)
        k = int(n * dup_frac)
        if k:
            v[rng.integers(0, n, size=k)] = v[rng.integers(0, n, size=k)]          # bitwise duplicates
        return v
    pos = values(n_pos, 3)
    idx = rng.integers(0, n_pos, size=(n_faces, 3)).astype(np.uint32)
    # stitch some proper fans in so that there are interior edges as well
    for f in range(0, n_faces - 2, 3):
        a, b, c, d = rng.choice(n_pos, size=4, replace=False)
        idx[f] = (a, b, c); idx[f + 1] = (a, c, d)
    m = dict(pos=pos, idx_pos=idx.reshape(-1))
    if with_uv:
        uv = values(max(3, n_pos + 7), 2); m["uv"] = uv; m["idx_uv"] = rng.integers(0, len(uv), size=3 * n_faces).astype(np.uint32)
    if with_nrm:
        nr = values(max(3, n_pos // 2), 3) - 0.5; nr /= np.maximum(1e-6, np.linalg.norm(nr, axis=1, keepdims=True)); nr = nr.astype(np.float32)
        m["nrm"] = nr; m["idx_nrm"] = rng.integers(0, len(nr), size=3 * n_faces).astype(np.uint32)
    return m
