"""Development check (no GPU): the geometry kernels through tests/hipemu against the oracle's bytes on a spread of meshes.
`python tools/dev_geom_check.py [quick|full]`; environment switches (UVOL_SIMT_W=..., UVOL_RELABEL=1, ...) select kernel forms as in
tests/test_hipemu_geom.py.  The oracle is the checker only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "universal-volumetric_amd")):
    sys.path.insert(0, p)
import numpy as np
import synth, uvol
import oracle as O

O.lib()
lib = os.path.join(ROOT, "tests", "hipemu", "libuvolcodec_hipemu.so")
mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
frames = [synth.torus_mesh(), synth.sphere_mesh(40, 21, charts=(5, 4)), synth.grid_mesh(), synth.sphere_mesh(24, 13, charts=(3, 2), crease=False)]
t = frames[0]
frames.append(dict(pos=t["pos"], idx_pos=t["idx_pos"]))
frames.append(dict(pos=np.concatenate([t["pos"], t["pos"][:1]]), idx_pos=np.concatenate([t["idx_pos"], np.array([0, len(t["pos"]), 5], np.uint32)])))
frames += list(synth.edge_case_meshes().values())
frames += [synth.random_soup_mesh(5), synth.random_soup_mesh(6, 60, 300), synth.shuffle_mesh(frames[1], seed=3)]
if mode == "full":
    frames += [synth.random_soup_mesh(s, 30 + 7 * s, 100 + 31 * s) for s in range(7, 27)]
    frames += synth.distinct_meshes(3, n_seg=60, n_ring=37, charts=(6, 5))
kw = {}
for k in ("Q_POSITION_ATTR", "Q_TEXTURE_ATTR", "Q_NORMAL_ATTR"):
    if os.environ.get(k):
        kw[k] = int(os.environ[k])
c = uvol.Codec(lib_path=lib, **kw)
t0 = time.time()
res = c.encode_mesh_batch(frames, raise_on_error=False)
# clean frames only (the compact layout holds), then the mixed batch again (the context remembers), then clean again
clean = [frames[0], frames[1], frames[3]]
res2 = c.encode_mesh_batch(clean) + c.encode_mesh_batch(frames, raise_on_error=False) + c.encode_mesh_batch(clean)
assert res2 == [res[0], res[1], res[3]] + res + [res[0], res[1], res[3]], "results depend on the layout mode"
bad = 0
for i, (f, r) in enumerate(zip(frames, res)):
    e = O.drc_encode(f["pos"], f["idx_pos"], f.get("uv"), f.get("idx_uv"), f.get("nrm"), f.get("idx_nrm"),
                     **({"qp": kw.get("Q_POSITION_ATTR", 11), "qt": kw.get("Q_TEXTURE_ATTR", 10), "qn": kw.get("Q_NORMAL_ATTR", 8)} if kw else {}))
    if r != e:
        bad += 1
        n = min(len(r or b""), len(e))
        first = next((k for k in range(n) if r[k] != e[k]), n) if r else -1
        print("frame %d: MISMATCH (got %s bytes, want %d, first difference at %d; %d faces)" % (i, len(r) if r else None, len(e), first, len(f["idx_pos"]) // 3))
print("%d frames, %d mismatches, %.1f s" % (len(frames), bad, time.time() - t0))
c.close()
sys.exit(1 if bad else 0)
