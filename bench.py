#!/usr/bin/env python3
"""bench.py — frames/s encode of 100k-vertex meshes + 2048^2 textures (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic frames resident in HBM:
FRAMES_PER_STEP geometry frames (one `uvol_encode_mesh_batch_dev` call = what FRAMES_PER_STEP
`draco_encoder` processes do, scripts/Encoder.py:256-267) and FRAMES_PER_STEP / KTX2_BATCH_SIZE
texture segments (one batched `uvol_encode_texture_segments_dev` call = that many `basisu` processes,
scripts/Encoder.py:279-298), geometry and texture on two HIP streams of the same GPU.  The timed
region ends when every .drc / .ktx2 byte is in host memory.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Frames shard one-per-GPU-batch (weak scaling: per-GPU work fixed); the only collective is the final
32-byte-per-rank manifest gather, over RCCL.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "universal-volumetric_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
I8_PEAK_TOPS = 3944.0        # dense i8 MFMA (16x16x64), MI355X_MICROARCH.md
PMC_FILE = "r02_pmc_traffic.json"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames-per-step", type=int, default=2160, help="frames in flight per step (per GPU); a frame holds ~75 MB of geometry workspace + 27 MB of inputs")
    ap.add_argument("--tex-size", type=int, default=2048)
    ap.add_argument("--segs", type=int, default=400, help="sphere segments (400 x 251 rings = 100,002 vertices)")
    ap.add_argument("--rings", type=int, default=251)
    ap.add_argument("--batch", type=int, default=5, help="KTX2_BATCH_SIZE")
    ap.add_argument("--distinct", type=int, default=5, help="distinct synthetic frames whose content is cycled")
    ap.add_argument("--shared-inputs", action="store_true", help="DIAGNOSTIC: all frames read the same --distinct input buffers / one texture segment (as before r01_k)")
    ap.add_argument("--geo-streams", type=int, default=1, help="geometry contexts (HIP streams); frames of a step are split evenly between them")
    ap.add_argument("--mesh-order", choices=["lattice", "shuffled"], default="lattice", help="DIAGNOSTIC 'shuffled': seeded permutation of faces and values (scan-like storage order, no cache-line locality between consecutive faces)")
    ap.add_argument("--tex-streams", type=int, default=1, help="texture contexts (HIP streams) fed by host threads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traverse-vbits-l2", type=int, default=0, help="LDS walkers (small batches): 1 = attribute traversers keep their vertex bitmap in L2")
    ap.add_argument("--tex-priority", type=int, default=1, help="1: texture contexts use a high-priority HIP stream")
    ap.add_argument("--geo-priority", type=int, default=0, help="DIAGNOSTIC: geometry contexts on high-priority streams too")
    ap.add_argument("--lockstep", action="store_true", help="barrier between all streams after every pass (default: each stream runs its passes back to back)")
    ap.add_argument("--tex-delay-ms", type=float, default=0.0, help="DIAGNOSTIC: the texture streams start each pass (--lockstep) / their first pass this long after the geometry streams")
    ap.add_argument("--geo-stagger-ms", type=float, default=0.0, help="one-time start delay of geometry stream g: g * this")
    ap.add_argument("--only", choices=["geo", "tex"], default=None, help="diagnostic: run only one half of the path (never the headline value)")
    ap.add_argument("--tex-cu", default="", help="MOD:MASKHEX: texture streams restricted to the CUs with (index mod MOD) in MASK; the geometry streams keep all CUs")
    ap.add_argument("--cu-split", type=int, default=0, help="16*G+T: CU residue masks (mod 4) for the geometry / texture streams, e.g. 0x7*16+0x8 = 120")
    ap.add_argument("--host-inputs", action="store_true", help="PCIe-inclusive variant: hand the C ABI host buffers (never the headline value)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import uvol, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    F = args.frames_per_step
    B = args.batch
    assert F % B == 0
    nseg = F // B

    # ---- synthetic frames (seeded), uploaded once; inputs are resident in HBM when timing starts ----
    meshes_h = [synth.sphere_mesh(args.segs, args.rings, frame=k, seed=k) for k in range(args.distinct)]
    if args.mesh_order == "shuffled":
        meshes_h = [synth.shuffle_mesh(m, seed=100 + k) for k, m in enumerate(meshes_h)]
    tex_h = synth.texture_sequence(B, size=args.tex_size, seed=0)
    V, Fc = len(meshes_h[0]["pos"]), len(meshes_h[0]["idx_pos"]) // 3
    keep = []
    dev_meshes = []
    # every frame of the step has its OWN input buffers in HBM (content cycles through the --distinct synthetic frames): no
    # frame finds its inputs in a cache because another frame of the batch read the same addresses
    base_t = [{k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in m.items()} for m in meshes_h]
    for i in range(F if not args.shared_inputs else args.distinct):
        m = meshes_h[i % args.distinct]
        t = base_t[i % args.distinct] if i < args.distinct else {k: v.clone() for k, v in base_t[i % args.distinct].items()}
        keep.append(t)
        mm = uvol.Mesh()
        mm.pos = t["pos"].data_ptr(); mm.n_pos = len(m["pos"]); mm.uv = t["uv"].data_ptr(); mm.n_uv = len(m["uv"])
        mm.nrm = t["nrm"].data_ptr(); mm.n_nrm = len(m["nrm"])
        mm.idx_pos = t["idx_pos"].data_ptr(); mm.idx_uv = t["idx_uv"].data_ptr(); mm.idx_nrm = t["idx_nrm"].data_ptr()
        mm.n_faces = len(m["idx_pos"]) // 3
        dev_meshes.append(mm)
    tex_d = [torch.from_numpy(a).to(dev) for a in tex_h]
    tex_ptrs = [t.data_ptr() for t in tex_d]
    # texture segments: own buffers AND own content per segment (the base segment shifted by whole 4x4 blocks along x)
    tex_seg_ptrs = []
    for s_ in range(nseg):
        if args.shared_inputs or s_ == 0:
            tex_seg_ptrs += tex_ptrs
        else:
            seg = [torch.roll(t, shifts=(4 * s_) % args.tex_size, dims=1).contiguous() for t in tex_d]
            keep.append(seg); tex_seg_ptrs += [t.data_ptr() for t in seg]
    torch.cuda.synchronize()

    cfg = dict(Q_POSITION_ATTR=11, Q_TEXTURE_ATTR=10, Q_NORMAL_ATTR=8, DRACO_COMPRESSION_LEVEL=7, KTX2_BATCH_SIZE=B, max_batch=F)
    gcfg, tcfg = dict(cfg), dict(cfg)
    GS = max(1, args.geo_streams)
    if args.tex_priority:
        tcfg.update(stream_priority=1)
    if args.geo_priority:
        gcfg.update(stream_priority=1)
    if args.traverse_vbits_l2 == 1:
        gcfg.update(traverse_vbits_l2=1)
    if args.tex_cu:            # --tex-cu MOD:MASK: the texture streams only run on CUs whose index modulo MOD has its bit set in MASK (hex)
        m_, r_ = args.tex_cu.split(":"); tcfg.update(cu_mod=int(m_), cu_residues=int(r_, 16))
    if args.cu_split:          # --cu-split GT: geometry on residues G (bitmask) of every 4 CUs, texture on residues T (hipExtStreamCreateWithCUMask)
        gcfg.update(cu_mod=4, cu_residues=int(args.cu_split) // 16); tcfg.update(cu_mod=4, cu_residues=int(args.cu_split) % 16)
    geos = [uvol.Codec(device=local_rank, **gcfg) for _ in range(GS)]
    texs = [uvol.Codec(device=local_rank, **tcfg) for _ in range(max(1, args.tex_streams))]
    out = {}

    host_frames = [meshes_h[i % args.distinct] for i in range(F)]

    gsl = [(gi * F // GS, (gi + 1) * F // GS) for gi in range(GS)]
    gbatches = [(uvol.Mesh * (b - a))(*[dev_meshes[i % len(dev_meshes)] for i in range(a, b)]) for a, b in gsl]

    def run_geo(gi):
        a, b = gsl[gi]
        out["drc_%d" % gi] = geos[gi].encode_mesh_batch(host_frames[a:b]) if args.host_inputs else geos[gi].encode_mesh_batch_dev(gbatches[gi], views=True)

    def run_tex(ti):
        mine = len(range(ti, nseg, len(texs)))            # segments of this step handled by texture context ti, ONE batched call
        if args.host_inputs:
            out["ktx2_%d" % ti] = texs[ti].encode_texture_segments([tex_h] * mine) if mine else []
        else:
            segs = range(ti, nseg, len(texs))
            out["ktx2_%d" % ti] = texs[ti].encode_texture_segments_dev([p_ for s_ in segs for p_ in tex_seg_ptrs[s_ * B:(s_ + 1) * B]], B, args.tex_size, args.tex_size) if mine else []

    def steps(k):
        """k passes over the batch.  Every stream (geometry sub-batch / texture share) runs its k passes back to back on its
        own host thread; with --lockstep all streams meet at a barrier after every pass instead.  --geo-stagger-ms delays
        geometry stream g by g * that much once, so that the streams' serial phases interleave instead of colliding."""
        errors = []

        def guarded(fn, *a):
            try:
                fn(*a)
            except BaseException as e:          # a failed stream must fail the whole bench, not shrink the work silently
                errors.append(e)

        def loop1(fn, arg, delay):
            if delay > 0:
                time.sleep(delay)
            fn(arg)

        def loop(fn, arg, delay):
            if delay > 0:
                time.sleep(delay)
            for _ in range(k):
                fn(arg)
        if args.lockstep:
            for _ in range(k):
                th = [threading.Thread(target=guarded, args=(run_geo, gi)) for gi in range(GS) if args.only != "tex"] + \
                     [threading.Thread(target=guarded, args=(loop1, run_tex, ti, args.tex_delay_ms * 1e-3)) for ti in range(len(texs)) if args.only != "geo"]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
        else:
            th = [threading.Thread(target=guarded, args=(loop, run_geo, gi, gi * args.geo_stagger_ms * 1e-3)) for gi in range(GS) if args.only != "tex"] + \
                 [threading.Thread(target=guarded, args=(loop, run_tex, ti, args.tex_delay_ms * 1e-3)) for ti in range(len(texs)) if args.only != "geo"]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errors:
            raise errors[0]
        out["ktx2"] = [k_ for ti in range(len(texs)) for k_ in out.get("ktx2_%d" % ti, [])]
        out["drc"] = [k_ for gi in range(GS) for k_ in out.get("drc_%d" % gi, [])]
        if args.only != "tex" and (len(out["drc"]) != F or not all(len(d_) for d_ in out["drc"])):
            raise RuntimeError("bench: %d of %d geometry frames encoded" % (len(out["drc"]), F))
        if args.only != "geo" and (len(out["ktx2"]) != nseg or not all(len(k_) for k_ in out["ktx2"])):
            raise RuntimeError("bench: %d of %d texture segments encoded" % (len(out["ktx2"]), nseg))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        steps(args.warmup)
    for g in geos:
        g.profile(True); g.profile_reset()
    for t in texs:
        t.profile(True); t.profile_reset()
    barrier()
    t0 = time.perf_counter()
    steps(args.steps)
    # manifest gather: {frames, segments, layers in last segment, bytes} per rank
    import shard
    nbytes = sum(len(x) for x in out["drc"]) + sum(len(x) for x in out["ktx2"])
    table = shard.gather_counts(F * args.steps, nseg * args.steps, B, nbytes, device=dev)      # RCCL all_gather when world > 1
    total_frames, total_segs, total_tex_frames, _ = shard.totals(table, B)
    assert total_frames == total_tex_frames or args.only                                                   # check_total_frames (Encoder.py:135)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])

    if rank == 0:
        drc_len = sum(len(x) for x in out["drc"]) / F
        ktx_len = sum(len(x) for x in out["ktx2"]) / F
        algo_per_frame = 32.0 * V + 12.0 * Fc + 4.0 * args.tex_size ** 2 + drc_len + ktx_len
        groups = []
        tg = {}
        for t in geos + texs:
            for g in t.profile_report():
                a = tg.setdefault(g["name"], dict(name=g["name"], launches=0, total_ms=0.0, algo_bytes=0))
                a["launches"] += g["launches"]; a["total_ms"] += g["total_ms"]; a["algo_bytes"] += g["algo_bytes"]
        groups += list(tg.values())
        groups.sort(key=lambda g: -g["total_ms"])
        mfma_grp = tg.get("tex.k10_sel_assign")                   # sub-scope of tex.k10_selector_codebook; its work field counts integer ops, not bytes
        # dominant kernel: the longest group of the CRITICAL PATH, i.e. of the geometry stream (its groups sum to ~95 % of a step;
        # the texture stream runs beside it and is idle a third of the time) - the texture selector-codebook group is about as
        # long but is ~140 launches of ten kernels, not a kernel.  With --only tex the longest texture group stands in.
        cand = [g for g in groups if g["name"].startswith("geo.")] or [g for g in groups if g["name"] != "tex.k10_sel_assign"]
        dom = cand[0]
        units = F // GS if dom["name"].startswith("geo.") else F // len(texs)          # frames one launch of that group processes
        avg_ms = dom["total_ms"] / max(1, dom["launches"])
        achieved = algo_per_frame * units / (avg_ms * 1e-3) / 1e9
        res = {
            "metric": "frames/s encode, 100k-vert mesh + 2048^2 texture",
            "value": total_frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (f32 only in the quantiser)", "data": "synthetic" + (" (host buffers, PCIe-inclusive)" if args.host_inputs else "") + (" DIAGNOSTIC %s only" % args.only if args.only else ""),
            "config": {"workload": "BASELINE configs[2] shape: %d-vertex/%d-face meshes + %dx%d RGBA8 ETC1S video segments of %d layers, "
                                   "%d frames per step, qp11/qt10/qn8/cl7; %s%s" % (V, Fc, args.tex_size, args.tex_size, B, F, "DIAGNOSTIC shuffled face / value order; " if args.mesh_order == "shuffled" else "",
                                   "DIAGNOSTIC shared input buffers" if args.shared_inputs else "every frame / segment reads its own input buffers in HBM"),
                       "frames_per_step": F, "ktx2_batch_size": B, "parallelism": "frames sharded per GPU; per GPU %d geometry + %d texture streams" % (GS, len(texs)),
                       "drc_bytes_per_frame": drc_len, "ktx2_bytes_per_frame": ktx_len, "mesh_order": args.mesh_order,
                       "geometry_workspace_bytes_per_frame": geos[0].mesh_workspace(**meshes_h[0])},
            "roofline": {"bound": "hbm", "kernel": dom["name"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom["name"], units), "avg_launch_ms": avg_ms, "units_per_launch": units,
                         "algorithmic_bytes_per_frame": algo_per_frame,
                         "end_to_end_achieved": algo_per_frame * total_frames / world / dt / 1e9},
            "kernel_groups_ms_per_step": {g["name"]: g["total_ms"] / args.steps for g in groups},
        }
        if mfma_grp and mfma_grp["total_ms"] > 0:
            # the one contraction on the path (exact i8 nearest-codeword search) against the dense i8
            # matrix-core rate of MI355X_MICROARCH.md (16x16x64: >= 3944 TOP/s); secondary to `roofline`, which stays the dominant kernel
            tops = mfma_grp["algo_bytes"] / (mfma_grp["total_ms"] * 1e-3) / 1e12
            res["roofline_mfma"] = {"bound": "mfma", "kernel": "tex.k10_sel_assign", "achieved": tops, "peak": I8_PEAK_TOPS, "unit": "TOP/s",
                                    "frac": tops / I8_PEAK_TOPS, "avg_launch_ms": mfma_grp["total_ms"] / max(1, mfma_grp["launches"]),
                                    "ops_per_launch": mfma_grp["algo_bytes"] / max(1, mfma_grp["launches"]), "dtype": "i8 x i8 -> i32"}
        res["quality"] = quality_gates(geos[0], texs[0], out, meshes_h[0], tex_h, B) if not args.only else None
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(meshes_h[0], tex_h, B)
        print(json.dumps(res))
    for g in geos:
        g.close()
    for t in texs:
        t.close()
    if world > 1:
        dist.destroy_process_group()


def quality_gates(geo, tex, out, mesh0, tex0, B):
    """SURVEY 8(d) 'quality gates reported with the speed', outside the timed region: frame 0 / segment 0 of the timed outputs are
    decoded on the GPU (this codec's decode path, itself bit-exact against the fixture-pinned decoders in the tests).  Geometry,
    independent of vertex order: the largest distance from an input position to the nearest decoded position against half a
    quantisation step per axis; texture: RGB PSNR of the decoded layers against the source (stored bottom-up, -y_flip) and
    bits per texel of the segment."""
    try:
        import numpy as np
        from scipy.spatial import cKDTree
        d = geo.decode_mesh_batch([bytes(out["drc"][0])])[0]
        pos = np.asarray(mesh0["pos"], np.float64)
        step = float((pos.max(0) - pos.min(0)).max()) / (2 ** 11 - 1)
        dist, _ = cKDTree(np.asarray(d["pos"], np.float64)).query(pos)
        dec = tex.decode_texture_segments([out["ktx2"][0]])[0]
        src = np.stack([np.asarray(a)[::-1] for a in tex0]).astype(np.float64)
        mse = float(np.mean((src[..., :3] - dec[..., :3].astype(np.float64)) ** 2))
        return {"frame": 0, "max_pos_err": float(dist.max()), "pos_half_step_diagonal": step / 2 * 3 ** 0.5, "faces": int(d["n_faces"]),
                "texture_psnr_rgb_db": 10.0 * float(np.log10(255.0 ** 2 / max(mse, 1e-12))), "texture_bits_per_texel": 8.0 * len(out["ktx2"][0]) / (dec.shape[0] * dec.shape[1] * dec.shape[2])}
    except Exception as e:                                   # never lose the bench line over the side report
        print("quality gates not computed: %r" % (e,), file=sys.stderr)
        return None


def pmc_traffic(group, units):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE are
    collected in their own runs, profiles/<PMC_FILE> records the per-frame figure and how it was corrected)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        e = t["kernels"].get(group)
        return None if e is None else e["hbm_bytes_per_frame"] * units
    except Exception:
        return None


def cpu_baseline(mesh, tex, B):
    """The CPU oracle (single-thread restatement, kind 'port') timed on this box's host cores on a bounded
    sample of the same workload.  The stock draco_encoder / basisu binaries are not in this image."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.lib()
    t = time.perf_counter()
    for _ in range(2):
        O.drc_encode(mesh["pos"], mesh["idx_pos"], mesh["uv"], mesh["idx_uv"], mesh["nrm"], mesh["idx_nrm"])
    t_geo = (time.perf_counter() - t) / 2
    t = time.perf_counter()
    O.ktx2_encode(tex)
    t_tex = time.perf_counter() - t
    fps = B / (B * t_geo + t_tex)
    res = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "2 geometry frames (%.3f s each) + 1 texture segment of %d layers (%.2f s), serial like scripts/Encoder.py" % (t_geo, B, t_tex),
           "host_cores_available": os.cpu_count()}
    # the same port on several host cores at once (basisu itself is multithreaded): P processes, each one texture
    # segment + its B geometry frames; bounded to one round (~ the serial sample's duration)
    try:
        import multiprocessing as mp
        P = max(1, min(16, (os.cpu_count() or 1) // 2))
        ctx = mp.get_context("fork")
        t = time.perf_counter()
        with ctx.Pool(P) as pool:
            pool.map(_cpu_segment, [(mesh, tex, B)] * P)
        dt = time.perf_counter() - t
        res["parallel"] = {"value": P * B / dt, "unit": "frames/s", "cores": P, "sample": "%d processes x (1 texture segment + %d geometry frames) in %.2f s" % (P, B, dt)}
    except Exception as e:                                   # the single-core figure above stands on its own
        res["parallel"] = {"error": repr(e)}
    return res


def _cpu_segment(a):
    mesh, tex, B = a
    import oracle as O
    for _ in range(B):
        O.drc_encode(mesh["pos"], mesh["idx_pos"], mesh["uv"], mesh["idx_uv"], mesh["nrm"], mesh["idx_nrm"])
    O.ktx2_encode(tex)
    return 0


if __name__ == "__main__":
    main()
