#!/usr/bin/env python3
"""bench.py — frames/s encode of 100k-vertex meshes + 2048^2 textures (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic frames resident in HBM:
FRAMES_PER_STEP geometry frames (one `uvol_encode_mesh_batch_dev` call = what FRAMES_PER_STEP
`draco_encoder` processes do, scripts/Encoder.py:256-267) and FRAMES_PER_STEP / KTX2_BATCH_SIZE
texture segments (one batched `uvol_encode_texture_segments_dev` call = that many `basisu` processes,
scripts/Encoder.py:279-298), geometry and texture on two HIP streams of the same GPU.  The timed
region ends when every .drc / .ktx2 byte is in host memory.

  python bench.py --gpus 1 --steps K --warmup W                      # weak scaling: every rank encodes --frames-per-step frames
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --gpus N --total-frames 1200 ...                    # STRONG scaling (BASELINE configs[3]): ONE job of that many
                                                                      # frames, split over the ranks by shard.plan; a step = the job

`value` is quoted with the inputs resident in HBM when the timed region starts - the bench contract's rule for this tier ("if the boundary hands over
host buffers, note the PCIe-inclusive rate - it is never `value`").  The SAME line carries, as top-level keys, what the SURVEY §8(d) boundary gives
(`survey_8d_boundary_frames_per_s`: pinned / pageable host memory through the blocking and the enqueue forms, and what the link allows for the bytes
the ABI takes) and BASELINE configs[2]'s own 300-frame job (`configs2_job_300_frames_per_s`).

`--gpus N` is the rank count: without a launcher `python bench.py --gpus N` starts its N ranks itself (torch.distributed.run on 127.0.0.1); under one,
WORLD_SIZE and the process group's size must equal it or the run exits non-zero.  The only collective is the final 32-byte-per-rank manifest gather
(SURVEY §8e), over RCCL.
On one GPU the line also carries `variants` (never `value`): the same path on the SURVEY §8(d) boundary (host buffers), as small
jobs (150 / 300 / 1200 frames -> the 8-GPU projection of configs[3]), on a scan-like storage order, and the decode path
(configs[4]); `--no-variants` skips them.
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
# One hardware queue per stream: the runtime's default is 4 hardware queues per process, and streams that share one run one after the
# other.  The codec's contexts hold two streams per geometry lane plus the texture stream (measured: 4 lanes, 2922 frames/s geometry-only
# on 4 queues, 3420 on 24).  Read by the HIP runtime when it initialises, i.e. before torch / the codec library are loaded.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
I8_PEAK_TOPS = 3944.0        # dense i8 MFMA (16x16x64), MI355X_MICROARCH.md
PMC_FILE = "r06_final_pmc_traffic.json"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames-per-step", type=int, default=2560, help="frames in flight per step (per GPU); a frame holds ~56 MB of geometry workspace + 27 MB of inputs: 2560 frames leave ~17 GB of the 288 GB free (2760 no longer fit)")
    ap.add_argument("--total-frames", type=int, default=0, help="STRONG scaling: one job of this many frames split over the ranks (shard.plan, whole texture segments per rank); "
                                                                "a step is the whole job.  BASELINE configs[3]: --total-frames 1200 --gpus 8")
    ap.add_argument("--strong-frames", type=int, default=1200, help="N > 1, weak form: the line also carries `strong_configs3`, ONE job of this many frames split over the ranks (BASELINE configs[3]: 1200)")
    ap.add_argument("--tex-size", type=int, default=2048)
    ap.add_argument("--segs", type=int, default=400, help="sphere segments (400 x 251 rings = 100,002 vertices)")
    ap.add_argument("--rings", type=int, default=251)
    ap.add_argument("--batch", type=int, default=5, help="KTX2_BATCH_SIZE")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic frames whose content is cycled (each its own tessellation; >= 64 for the headline)")
    ap.add_argument("--connectivity", choices=["distinct", "identical"], default="distinct",
                    help="distinct (headline): every synthetic frame has its own connectivity and vertex count, like a capture's frames; identical (DIAGNOSTIC): "
                         "the frames share one index array (an animated mesh of fixed topology) and the lane-per-walker kernels run in lock step")
    ap.add_argument("--parity-frames", type=int, default=8, help="frames of the timed step whose bytes are compared with the CPU oracle's after the timed region (0: skip)")
    ap.add_argument("--shared-inputs", action="store_true", help="DIAGNOSTIC: all frames read the same --distinct input buffers / one texture segment")
    ap.add_argument("--geo-streams", type=int, default=1, help="geometry contexts (HIP streams); frames of a step are split evenly between them")
    ap.add_argument("--mesh-order", choices=["lattice", "shuffled"], default="lattice", help="DIAGNOSTIC 'shuffled': seeded permutation of faces and values (scan-like storage order, no cache-line locality between consecutive faces)")
    ap.add_argument("--tex-streams", type=int, default=1, help="texture contexts (HIP streams) fed by host threads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the workload variants reported next to the headline on one GPU")
    ap.add_argument("--traverse-vbits-l2", type=int, default=0, help="LDS walkers (small batches): 1 = attribute traversers keep their vertex bitmap in L2")
    ap.add_argument("--tex-priority", type=int, default=1, help="1: texture contexts use a high-priority HIP stream")
    ap.add_argument("--geo-priority", type=int, default=0, help="1: geometry contexts use a high-priority HIP stream (diagnostic)")
    ap.add_argument("--geo-cus", default="", help="DIAGNOSTIC 'mod:residues' (residues in hex): the geometry contexts' streams run on the CUs whose mask index %% mod has its bit set (hipExtStreamCreateWithCUMask; mask bit i lies on XCD i %% 8: profiles/r05_xcd_census.json)")
    ap.add_argument("--tex-cus", default="", help="DIAGNOSTIC: the same for the texture contexts")
    ap.add_argument("--blocking-calls", action="store_true", help="DIAGNOSTIC: one blocking geometry call per pass instead of enqueued passes completed by one uvol_sync")
    ap.add_argument("--lockstep", action="store_true", help="barrier between all streams after every pass (default: each stream runs its passes back to back)")
    ap.add_argument("--only", choices=["geo", "tex"], default=None, help="diagnostic: run only one half of the path (never the headline value)")
    ap.add_argument("--host-inputs", action="store_true", help="PCIe-inclusive variant: hand the C ABI host buffers (never the headline value)")
    ap.add_argument("--host-pinned", action="store_true", help="with --host-inputs: the input arrays lie in uvol_host_alloc (page-locked) memory")
    ap.add_argument("--host-enqueued", action="store_true", help="with --host-inputs: the passes go through the enqueue forms (uvol_*_async on host buffers, one uvol_sync)")
    ap.add_argument("--tex-enqueued", type=int, default=0, help="1: the texture passes of a device-input job go through uvol_encode_texture_segments_dev_async + one uvol_sync (a pass's last two parts stay in flight beside the next pass's first); DIAGNOSTIC: measured 3534 - 3540 against 3463 - 3592 frames/s, inside the scatter - the texture context is not the critical path")
    ap.add_argument("--background", choices=["off", "h2d_16m", "h2d_256m", "d2d"], default="off",
                    help="DIAGNOSTIC (profiles/r06_uplink_forms.json): a host thread keeps copying during the timed steps - page-locked host memory to the device in copies of 16 MiB "
                         "(the runtime's blit kernel) or 256 MiB (its SDMA engines), or device to device at about the link's rate - to see what an upload does to the encoders beside it")
    args = ap.parse_args()

    # `--gpus N` IS the rank count.  Under a launcher (torch.distributed.run sets WORLD_SIZE) the two must agree; without one,
    # `python bench.py --gpus N` launches its N ranks itself - the same command form as N = 1.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s): the line would misreport n_gpus" % (args.gpus, world))
    one_dev = os.environ.get("UVOL_BENCH_ONE_DEVICE") == "1"      # DIAGNOSTIC: every rank drives device 0, collectives over gloo (the N > 1 code path on a 1-GPU box)
    launch_only = os.environ.get("UVOL_BENCH_LAUNCH_ONLY") == "1"  # TEST (no GPU needed): rendezvous + the manifest gather of an empty job, then one line

    import numpy as np
    import torch
    import uvol, synth, shard

    if one_dev:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if launch_only:
            dist.init_process_group(backend="gloo")
        elif not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
        elif one_dev:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="gloo")
        else:
            if torch.cuda.device_count() < world:
                raise SystemExit("bench.py: --gpus %d but this node shows %d device(s)" % (world, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:                 # what the collective library saw, not what the environment says
            raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s)" % (args.gpus, dist.get_world_size()))
    if launch_only:
        f_lo, nf, s_lo, ns = shard.plan(args.total_frames or 1200, args.batch, world, rank)
        table = shard.gather_counts(nf, ns, args.batch, 0, device=None)
        if rank == 0:
            print(json.dumps({"launch_only": True, "n_gpus": (dist.get_world_size() if world > 1 else 1), "ranks_gathered": int(table.shape[0]),
                              "frames_gathered": int(shard.totals(table, args.batch)[0])}))
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if one_dev else dev              # where the collectives' tensors live

    B = args.batch
    strong = args.total_frames > 0
    if strong:                     # this rank's block of the ONE job (contiguous whole segments, SURVEY 8e / scripts/Encoder.py:103-154 accounting)
        f_lo, F, s_lo, nseg = shard.plan(args.total_frames, B, world, rank)
        assert args.total_frames % B == 0, "--total-frames must be a multiple of KTX2_BATCH_SIZE in the bench"
    else:
        F = args.frames_per_step
        assert F % B == 0
        nseg = F // B
    variants_on = world == 1 and not args.no_variants and (not args.only or os.environ.get("UVOL_VARIANTS_WITH_ONLY") == "1") and not args.host_inputs and not strong and args.mesh_order == "lattice"
    F_alloc = max(F, 1)

    # ---- synthetic frames (seeded), uploaded once; inputs are resident in HBM when timing starts ----
    def identical_meshes():                # round 1-3's workload: five seeded deformations of ONE tessellation (shared index arrays)
        return [synth.sphere_mesh(args.segs, args.rings, frame=k, seed=k) for k in range(min(args.distinct, 5))]
    meshes_h = synth.distinct_meshes(args.distinct, args.segs, args.rings) if args.connectivity == "distinct" else identical_meshes()
    if args.mesh_order == "shuffled":
        meshes_h = [synth.shuffle_mesh(m, seed=100 + k) for k, m in enumerate(meshes_h)]
    tex_h = synth.texture_sequence(B, size=args.tex_size, seed=0)
    keep = []
    dev_meshes = []
    frame_t = []

    def build_inputs(mh):
        """Every frame of the step gets its OWN input buffers in HBM (content cycles through the synthetic frames `mh`): no frame finds
        its inputs in a cache because another frame of the batch read the same addresses.  Replaces the previous set."""
        del dev_meshes[:]; del frame_t[:]
        torch.cuda.empty_cache()
        nd = len(mh)
        base_t = [{k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in m.items()} for m in mh]
        for i in range(F_alloc if not args.shared_inputs else nd):
            m = mh[i % nd]
            t = base_t[i % nd] if i < nd else {k: v.clone() for k, v in base_t[i % nd].items()}
            frame_t.append(t)
            mm = uvol.Mesh()
            mm.pos = t["pos"].data_ptr(); mm.n_pos = len(m["pos"]); mm.uv = t["uv"].data_ptr(); mm.n_uv = len(m["uv"])
            mm.nrm = t["nrm"].data_ptr(); mm.n_nrm = len(m["nrm"])
            mm.idx_pos = t["idx_pos"].data_ptr(); mm.idx_uv = t["idx_uv"].data_ptr(); mm.idx_nrm = t["idx_nrm"].data_ptr()
            mm.n_faces = len(m["idx_pos"]) // 3
            dev_meshes.append(mm)
        torch.cuda.synchronize()
    if not args.host_inputs:                                   # (--host-inputs: the frames stay in host memory, nothing is resident on the device)
        build_inputs(meshes_h)
    ND = len(meshes_h)
    V = sum(len(meshes_h[i % ND]["pos"]) for i in range(F_alloc)) / F_alloc                  # per-frame averages over the frames of a step
    Fc = sum(len(meshes_h[i % ND]["idx_pos"]) // 3 for i in range(F_alloc)) / F_alloc
    tex_d = [torch.from_numpy(a).to(dev) for a in tex_h]
    tex_ptrs = [t.data_ptr() for t in tex_d]
    # texture segments: own buffers AND own content per segment (the base segment shifted by whole 4x4 blocks along x)
    tex_seg_ptrs = []
    for s_ in range(max(nseg, 1) if not args.host_inputs else 1):
        if args.shared_inputs or s_ == 0:
            tex_seg_ptrs += tex_ptrs
        else:
            seg = [torch.roll(t, shifts=(4 * s_) % args.tex_size, dims=1).contiguous() for t in tex_d]
            keep.append(seg); tex_seg_ptrs += [t.data_ptr() for t in seg]
    torch.cuda.synchronize()

    cfg = dict(Q_POSITION_ATTR=11, Q_TEXTURE_ATTR=10, Q_NORMAL_ATTR=8, DRACO_COMPRESSION_LEVEL=7, KTX2_BATCH_SIZE=B, max_batch=max(F_alloc, 1))
    gcfg, tcfg = dict(cfg), dict(cfg)
    GS = max(1, args.geo_streams)
    if args.tex_priority:
        tcfg.update(stream_priority=1)
    if args.geo_priority:
        gcfg.update(stream_priority=1)
    for spec, c_ in ((args.geo_cus, gcfg), (args.tex_cus, tcfg)):
        if spec:
            if spec.startswith("xcd:"):                          # "xcd:lo-hi": the CUs with ordinal lo .. hi - 1 inside every XCD
                lo_, hi_ = (int(x_) for x_ in spec[4:].split("-")); c_.update(cu_mod=-1, cu_residues=(lo_ << 8) | hi_)
            else:
                c_.update(cu_mod=int(spec.split(":")[0]), cu_residues=int(spec.split(":")[1], 16))
    if args.traverse_vbits_l2 == 1:
        gcfg.update(traverse_vbits_l2=1)
    geos = [uvol.Codec(device=local_rank, **gcfg) for _ in range(GS)]
    texs = [uvol.Codec(device=local_rank, **tcfg) for _ in range(max(1, args.tex_streams))]
    out = {}

    _pin = {}

    def pinned_inputs():
        """The synthetic frames and the texture layers once more, in page-locked host memory (uvol_host_alloc)."""
        if not _pin:
            need = sum(v.nbytes + 256 for m in meshes_h for v in m.values()) + sum(t.nbytes + 256 for t in tex_h) + (1 << 20)
            ar = uvol.PinnedArena(need)
            _pin["arena"] = ar
            _pin["m"] = [{k: ar.put(v) for k, v in m.items()} for m in meshes_h]
            _pin["t"] = [ar.put(t) for t in tex_h]
        return _pin["m"], _pin["t"]

    class Job:
        """One pass over the first n frames / n // B segments of the resident inputs (n = F for the headline; the variants run
        smaller jobs on the same contexts and buffers)."""
        def __init__(self, n, host=False, only=args.only, blocking=False, host_enqueued=False, pinned=False):
            self.n, self.nseg, self.host, self.only, self.blocking = n, n // B, host, only, blocking or args.blocking_calls
            self.tex_h = tex_h
            if host and pinned:                                  # the caller's arrays in uvol_host_alloc memory: no staging copy inside the library
                pm, pt = pinned_inputs()
                self.tex_h = pt
            self.host_enqueued = host and host_enqueued         # host inputs through the enqueue forms (uvol_*_async on host buffers + uvol_sync)
            self.gsl = [(gi * n // GS, (gi + 1) * n // GS) for gi in range(GS)]
            nd = len(dev_meshes)
            self.gb = None if host else [(uvol.Mesh * (b - a))(*[dev_meshes[i % nd] for i in range(a, b)]) for a, b in self.gsl]
            self.hf = [(pm if (host and pinned) else meshes_h)[i % ND] for i in range(n)] if host else None

        def run_geo_passes(self, gi, kk):
            """kk passes of this geometry stream.  Device inputs: the passes are ENQUEUED (uvol_encode_mesh_batch_dev_async) and completed
            by one uvol_sync - the context then runs the front end of pass k + 1 beside the walkers of pass k (two output buffer sets in
            turn: a pass's bytes are in host memory when the pass after the next one starts).  --blocking-calls: one blocking call per pass."""
            a, b = self.gsl[gi]
            if self.host_enqueued and b > a:
                for k in range(kk):
                    geos[gi].start_mesh_batch(self.hf[a:b], slot=k & 1)
                res = geos[gi].finish()
                if any(r is None for r in res[-1]):
                    raise RuntimeError("bench: a geometry frame failed")
                out["drc_%d" % gi] = res[-1]
                return
            if self.host or self.blocking or b <= a:
                for _ in range(kk):
                    self.run_geo(gi)
                return
            for k in range(kk):
                geos[gi].start_mesh_batch_dev(self.gb[gi], slot=k & 1)
            res = geos[gi].finish()
            if any(r is None for r in res[-1]):
                raise RuntimeError("bench: a geometry frame failed")
            out["drc_%d" % gi] = res[-1]

        def run_geo(self, gi):
            a, b = self.gsl[gi]
            if b > a:
                out["drc_%d" % gi] = geos[gi].encode_mesh_batch(self.hf[a:b], views=True) if self.host else geos[gi].encode_mesh_batch_dev(self.gb[gi], views=True)
            else:
                out["drc_%d" % gi] = []

        def run_tex(self, ti):
            segs = range(ti, self.nseg, len(texs))           # segments of this step handled by texture context ti, ONE batched call
            if not len(segs):
                out["ktx2_%d" % ti] = []
            elif self.host:
                out["ktx2_%d" % ti] = texs[ti].encode_texture_segments([self.tex_h] * len(segs), views=True)
            else:
                out["ktx2_%d" % ti] = texs[ti].encode_texture_segments_dev([p_ for s_ in segs for p_ in tex_seg_ptrs[s_ * B:(s_ + 1) * B]], B, args.tex_size, args.tex_size, views=True)

        def steps(self, k):
            """k passes.  Every stream (geometry sub-batch / texture share) runs its k passes back to back on its own host
            thread; with --lockstep all streams meet at a barrier after every pass instead."""
            errors = []

            def guarded(fn, *a):
                try:
                    fn(*a)
                except BaseException as e:          # a failed stream must fail the whole bench, not shrink the work silently
                    errors.append(e)

            def loop(fn, arg, kk):
                if args.tex_enqueued and fn == self.run_tex and not self.host and not self.blocking:      # device inputs through the enqueue form
                    segs = range(arg, self.nseg, len(texs))
                    if len(segs):
                        pl = [p_ for s_ in segs for p_ in tex_seg_ptrs[s_ * B:(s_ + 1) * B]]
                        for k_ in range(kk):
                            texs[arg].start_texture_segments_dev(pl, B, args.tex_size, args.tex_size, slot=k_ & 1)
                        out["ktx2_%d" % arg] = texs[arg].finish()[-1]
                        return
                if self.host_enqueued and fn == self.run_tex:            # texture share of a host-input job, enqueued like the geometry share
                    segs = range(arg, self.nseg, len(texs))
                    if len(segs):
                        for k_ in range(kk):
                            texs[arg].start_texture_segments([self.tex_h] * len(segs), slot=k_ & 1)
                        out["ktx2_%d" % arg] = texs[arg].finish()[-1]
                        return
                for _ in range(kk):
                    fn(arg)
            rounds = [(1, k)] if not args.lockstep else [(k, 1)]
            for reps, kk in rounds:
                for _ in range(reps):
                    th = [threading.Thread(target=guarded, args=(self.run_geo_passes, gi, kk)) for gi in range(GS) if self.only != "tex"] + \
                         [threading.Thread(target=guarded, args=(loop, self.run_tex, ti, kk)) for ti in range(len(texs)) if self.only != "geo"]
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
            if errors:
                raise errors[0]
            out["ktx2"] = [k_ for ti in range(len(texs)) for k_ in out.get("ktx2_%d" % ti, [])]
            out["drc"] = [k_ for gi in range(GS) for k_ in out.get("drc_%d" % gi, [])]
            if self.only != "tex" and (len(out["drc"]) != self.n or not all(len(d_) for d_ in out["drc"])):
                raise RuntimeError("bench: %d of %d geometry frames encoded" % (len(out["drc"]), self.n))
            if self.only != "geo" and (len(out["ktx2"]) != self.nseg or not all(len(k_) for k_ in out["ktx2"])):
                raise RuntimeError("bench: %d of %d texture segments encoded" % (len(out["ktx2"]), self.nseg))

        def timed(self, k, w=1):
            """frames/s of k passes after w warm-up passes (single GPU, used by the variants)."""
            if w:
                self.steps(w)
            torch.cuda.synchronize(); t = time.perf_counter()
            self.steps(k)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            return {"frames_per_s": self.n * k / dt, "ms_per_pass": 1000.0 * dt / k, "frames": self.n, "passes": k}

    def set_profiling(on):
        for c in geos + texs:
            c.profile(on)
            if on:
                c.profile_reset()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    main_job = Job(F, host=args.host_inputs, host_enqueued=args.host_enqueued, pinned=args.host_pinned)
    if args.warmup:
        main_job.steps(args.warmup)
    bg = {"stop": False, "bytes": 0, "t": 0.0}
    if args.background != "off":
        nb = (256 << 20) if args.background == "h2d_256m" else (16 << 20)
        b_src = torch.empty(nb, dtype=torch.uint8).pin_memory() if args.background != "d2d" else torch.empty(nb, dtype=torch.uint8, device=dev)
        b_dst = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(4)]
        b_st = torch.cuda.Stream(device=dev)

        def bg_loop():
            t_ = time.perf_counter(); k_ = 0
            while not bg["stop"]:
                with torch.cuda.stream(b_st):
                    for _ in range(4):
                        b_dst[k_ & 3].copy_(b_src, non_blocking=True); k_ += 1
                b_st.synchronize()
                if args.background == "d2d":                    # paced to ~55 GB/s: 64 MiB every 1.2 ms
                    time.sleep(max(0.0, k_ * nb / 55e9 - (time.perf_counter() - t_)))
            bg["bytes"] = k_ * nb; bg["t"] = time.perf_counter() - t_
        bg_th = threading.Thread(target=bg_loop); bg_th.start()
    set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    main_job.steps(args.steps)
    # manifest gather: {frames, segments, layers in last segment, bytes} per rank
    nbytes = sum(len(x) for x in out["drc"]) + sum(len(x) for x in out["ktx2"])
    table = shard.gather_counts(F * args.steps, nseg * args.steps, B, nbytes, device=cdev)      # RCCL all_gather when world > 1
    total_frames, total_segs, total_tex_frames, _ = shard.totals(table, B)
    assert total_frames == total_tex_frames or args.only                                                   # check_total_frames (Encoder.py:135)
    if strong:
        assert total_frames == args.total_frames * args.steps, (total_frames, args.total_frames)
    barrier()
    dt = time.perf_counter() - t0
    if args.background != "off":
        bg["stop"] = True; bg_th.join()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])

    drc_len = sum(len(x) for x in out["drc"]) / max(F, 1)
    ktx_len = sum(len(x) for x in out["ktx2"]) / max(F, 1)
    reports = [t.profile_report() for t in geos + texs]        # (of the timed steps: what follows runs without the event brackets)
    # N > 1: next to the throughput curve (`value`: every GPU its own stream of passes, weak scaling) the STRONG job of BASELINE configs[3] -
    # ONE 1200-frame sequence split over the ranks by whole texture segments, each rank's share one blocking job, the 32-byte manifest
    # gather inside the timed region - so that one driver line per N carries both curves
    strong_extra = None
    if world > 1 and not strong and not args.host_inputs and not args.only:
        TF = args.strong_frames
        _, Fs, _, nsegs = shard.plan(TF, B, world, rank)
        if 0 < Fs <= F:
            set_profiling(False)
            js = Job(Fs, blocking=True)
            js.steps(1)
            barrier(); ts = time.perf_counter()
            js.steps(2)
            tbl = shard.gather_counts(Fs * 2, nsegs * 2, B, 0, device=cdev)
            barrier(); ds = time.perf_counter() - ts
            tt = torch.tensor([ds], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ds = float(tt[0])
            strong_extra = {"what": "BASELINE configs[3]: ONE job of %d frames split over %d ranks by whole texture segments (rank 0: %d frames), 2 jobs timed after 1 warm-up, max over ranks" % (TF, world, Fs),
                            "frames_per_s": 2.0 * TF / ds, "ms_per_job": 500.0 * ds, "scaling": "strong", "frames_gathered": int(shard.totals(tbl, B)[0])}
            set_profiling(True)

    failed = 0
    if rank == 0:
        algo_per_frame = 32.0 * V + 12.0 * Fc + 4.0 * args.tex_size ** 2 + drc_len + ktx_len
        tg = {}
        for rep in reports:
            for g in rep:
                a = tg.setdefault(g["name"], dict(name=g["name"], launches=0, total_ms=0.0, algo_bytes=0))
                a["launches"] += g["launches"]; a["total_ms"] += g["total_ms"]; a["algo_bytes"] += g["algo_bytes"]
        groups = sorted(tg.values(), key=lambda g: -g["total_ms"])
        mfma_grp = tg.get("tex.k10_sel_assign")                   # sub-scope of tex.k10_selector_codebook; its work field counts integer ops, not bytes
        # dominant kernel: the group with the largest total time over ALL groups of both contexts (only the sub-scope
        # tex.k10_sel_assign, which is counted inside tex.k10_selector_codebook, is not a candidate)
        cand = [g for g in groups if g["name"] != "tex.k10_sel_assign"]
        dom = cand[0]
        # frames one launch of that group processes: a geometry call is cut into groups on the context's lanes (each its own launch), a
        # texture call is one launch per stage
        units = max(1.0, (F if dom["name"].startswith("geo.") else nseg * B) * args.steps / max(1, dom["launches"]))
        avg_ms = dom["total_ms"] / max(1, dom["launches"])
        achieved = algo_per_frame * units / (avg_ms * 1e-3) / 1e9
        workload = ("BASELINE configs[3] shape: ONE job of %d frames split over %d rank(s) by whole texture segments (rank 0: %d frames)" % (args.total_frames, world, F)) if strong else \
                   ("BASELINE configs[2] shape, %d frames per step" % F)
        conn = ("distinct connectivity per frame (%d tessellations cycled: vertex / face counts, chart seams and quad diagonals differ from frame to frame)" % ND) if args.connectivity == "distinct" else \
               "DIAGNOSTIC identical connectivity (the frames share one index array: lane-per-walker kernels run in lock step)"
        res = {
            "metric": "frames/s encode, 100k-vert mesh + 2048^2 texture",
            "value": total_frames / dt, "unit": "frames/s", "n_gpus": (dist.get_world_size() if world > 1 else 1), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u8/int32 (f32 only in the quantiser)", "data": "synthetic" + (" (host buffers, PCIe-inclusive)" if args.host_inputs else "") + (" DIAGNOSTIC %s only" % args.only if args.only else ""),
            "config": {"workload": "%s, %s: ~%d-vertex/~%d-face meshes + %dx%d RGBA8 ETC1S video segments of %d layers, qp11/qt10/qn8/cl7; %s%s"
                                   % (workload, conn, round(V), round(Fc), args.tex_size, args.tex_size, B, "DIAGNOSTIC shuffled face / value order; " if args.mesh_order == "shuffled" else "",
                                      "DIAGNOSTIC shared input buffers" if args.shared_inputs else "every frame / segment reads its own input buffers in HBM"),
                       "frames_per_step": F if not strong else args.total_frames, "ktx2_batch_size": B,
                       "parallelism": "frames sharded per GPU in blocks of whole segments; per GPU %d geometry + %d texture streams" % (GS, len(texs)),
                       "drc_bytes_per_frame": drc_len, "ktx2_bytes_per_frame": ktx_len, "mesh_order": args.mesh_order, "connectivity": args.connectivity, "distinct_frames": ND,
                       "vertices_per_frame": V, "faces_per_frame": Fc,
                       "geometry_workspace_bytes_per_frame": geos[0].mesh_workspace(**meshes_h[0]),
                       "hbm_in_use_gb_after_timed_steps": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1)},
            "roofline": {"bound": "hbm", "kernel": dom["name"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom["name"], units), "avg_launch_ms": avg_ms, "units_per_launch": units,
                         "algorithmic_bytes_per_frame": algo_per_frame,
                         "launches_per_step": dom["launches"] / max(1, args.steps),
                         "frac_per_pass": achieved / HBM_PEAK_GBS * dom["launches"] / max(1, args.steps),
                         "note": "a pass is cut into groups on the context's lanes whose launches of this kernel run side by side: `frac` prices ONE launch (its frames, its duration) against the whole chip, as specified; "
                                 "frac x launches_per_step is what the concurrent launches move together while they overlap",
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
        if args.background != "off":
            res["background"] = {"kind": args.background, "GBps": bg["bytes"] / max(bg["t"], 1e-9) / 1e9}
        res["quality"] = quality_gates(geos[0], texs[0], out, meshes_h[0], tex_h, B) if not args.only and F else None
        if args.parity_frames > 0 and F and not args.host_inputs:
            res["parity"] = parity_check(out, meshes_h, tex_h, args.parity_frames, args.only, args.mesh_order)
            res["parity_checked_frames"] = res["parity"].get("geometry_frames_equal_to_oracle", 0)
        if strong_extra:
            res["strong_configs3"] = strong_extra
        if variants_on:
            try:
                args._geo_cfg = dict(gcfg, device=local_rank)
                res["variants"] = {}
                run_variants(res["variants"], args, Job, set_profiling, frame_t, meshes_h, tex_h, out, geos, texs, keep, dev, res["ms_per_step"], F, B, V, Fc, local_rank, build_inputs, identical_meshes, dev_meshes)
            except Exception as e:                                   # the headline stands on its own
                res["variants"]["error"] = repr(e)                  # (the variants measured before the fault stay in the line)
            # beside `value` (inputs resident in HBM, as the bench contract prescribes): the same workload on SURVEY 8(d)'s boundary and BASELINE configs[2]'s own job
            hi = res["variants"].get("host_inputs") or {}
            if "pinned_enqueued" in hi:
                res["survey_8d_boundary_frames_per_s"] = {"pinned_host_memory_enqueued": hi["pinned_enqueued"]["frames_per_s"], "pinned_host_memory_blocking": hi.get("pinned", {}).get("frames_per_s"),
                                                          "pageable_host_memory_blocking": hi.get("frames_per_s"), "pageable_host_memory_enqueued": hi.get("enqueued_passes"),
                                                          "link_bound": hi.get("link", {}).get("frames_per_s_the_link_allows")}
            if "job_300" in res["variants"]:
                res["configs2_job_300_frames_per_s"] = res["variants"]["job_300"]["frames_per_s"]
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(meshes_h[0], tex_h, B)
        print(json.dumps(res), flush=True)
        if (res.get("parity") or {}).get("mismatches"):          # a fast line whose bytes differ from the reference algorithm's is not a result
            failed = 3
    for c in geos + texs:
        c.close()
    if world > 1:
        dist.destroy_process_group()
    if failed:
        raise SystemExit(failed)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as the driver's own N > 1 command does
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>),
    one process per GPU, and hand their exit code back.  Rank 0 prints the one JSON line."""
    import socket
    import subprocess
    with socket.socket() as s_:                                # a free port on the loopback interface
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this host driver)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def run_variants(v, args, Job, set_profiling, frame_t, meshes_h, tex_h, out, geos, texs, keep, dev, ms_step, F, B, V, Fc, local_rank, build_inputs, identical_meshes, dev_meshes):
    """The same path on other boundaries / job sizes / storage orders, each a few passes (about 25 s in all), reported NEXT TO the
    headline: what a user of the reference sees depends on where the inputs are and how large the job is."""
    import numpy as np
    import torch
    import uvol, synth
    def note(what):                                            # progress on stderr (a fault in a variant is then attributable)
        print("[bench] variant: " + what, file=sys.stderr, flush=True)
        v["in_progress"] = what                                # (stays in the line only if this variant raises)
    def trim_geos(tex_too=False):                              # a lane's workspace only grows, and the variants below change its shape: back to the device
        for c in geos + (texs if tex_too else []):             # (uvol_trim; the contexts and their streams stay - fresh contexts measured 13 % slower,
            c.trim()                                           #  profiles/r04_d_bench.json against r04_e)
        torch.cuda.empty_cache()
    set_profiling(False)                                       # (the variants run without the per-group hipEvent brackets; an events-off copy of the headline said nothing - a 3-pass run amortises
                                                               #  the last groups' tail over fewer passes than the headline's 6 and reads 5 % lower with or without events: removed)
    # (1) small jobs (latency-bound: the serial walks of a frame do not shrink with the batch) and the 8-GPU projection of
    #     BASELINE configs[3] (1200 frames, 150 per GPU): 8 x rate(150-frame job) / rate(1200-frame job on one GPU)
    for n in (150, 300, 1200):
        if n <= F:
            note("job_%d" % n)
            v["job_%d" % n] = dict(Job(n, blocking=True).timed(3, 1), note="ONE job = one blocking call per pass (its latency is what the 8-GPU projection needs)")
    if "job_150" in v and "job_1200" in v:
        v["projected_8gpu_speedup_configs3"] = {"value": 8.0 * v["job_150"]["frames_per_s"] / v["job_1200"]["frames_per_s"],
                                                "how": "8 x frames/s of the 150-frame share of one GPU / frames/s of the whole 1200-frame job on one GPU; "
                                                       "the manifest gather (32 bytes per rank) is not modelled"}
    if 300 <= F:                                               # a STREAM of 300-frame jobs: enqueued calls, consecutive jobs overlap on the context's lanes
        note("stream_of_300_frame_jobs")
        v["stream_of_300_frame_jobs"] = dict(Job(300).timed(6, 2), note="300-frame jobs enqueued back to back (uvol_encode_mesh_batch_dev_async), one uvol_sync at the end")
    if args.only:                                              # (diagnostic: UVOL_VARIANTS_WITH_ONLY=1) one half of the path, job sizes only
        v.pop("in_progress", None)
        return v
    # (2) scan-like storage order: the resident input buffers are overwritten with a seeded permutation of faces and values
    note("shuffled_order")
    trim_geos()                                                # (relabelled frames use the general layout: the lanes' workspaces are re-sized, not stacked on the headline's)
    sh = [synth.shuffle_mesh(m, seed=100 + k) for k, m in enumerate(meshes_h)]
    for i, t in enumerate(frame_t):
        for key, val in sh[i % len(sh)].items():
            t[key].copy_(torch.from_numpy(np.ascontiguousarray(val)), non_blocking=False)
    torch.cuda.synchronize()
    v["shuffled_order"] = dict(Job(F).timed(2, 1), note="same surfaces, faces and value arrays in a seeded random order (no locality between consecutive faces)")
    # (2b) rounds 1-3's headline workload: the frames share ONE index array, walkers of a wave never diverge (new input buffers)
    if args.connectivity == "distinct":
        note("identical_connectivity")
        trim_geos()
        build_inputs(identical_meshes())
        v["identical_connectivity"] = dict(Job(F).timed(2, 1), note="DIAGNOSTIC: five deformations of one tessellation (shared index arrays): the lane-per-walker kernels move in lock step; "
                                                                     "this was `value` until round 3 and flatters the dominant kernel")
    # (2c) one blocking geometry call per pass (the whole call is one group on the context's first lane, which then holds a workspace of
    #      the full call: its workspaces are given back before and after; the headline's inputs are rebuilt)
    note("blocking_calls")
    if args.connectivity == "distinct":
        build_inputs(meshes_h)
    trim_geos()
    kb = max(2, min(args.steps, 4))
    v["blocking_calls"] = dict(Job(F, blocking=True).timed(kb, 1), note="headline workload, one blocking geometry call per pass instead of enqueued passes")
    v["blocking_calls"]["enqueued_same_passes"] = Job(F).timed(kb, 1)["frames_per_s"]      # (short runs flatter both: the texture context finishes its passes early)
    # (3) SURVEY 8(d) boundary: inputs in host memory -> bytes in host memory (PCIe inclusive); the device copies of the inputs go first
    note("host_inputs")
    frame_t.clear(); keep.clear(); del dev_meshes[:]
    trim_geos(tex_too=True)                                    # (the host path cuts a call into more groups than the device path)
    nh = F                                                     # (1080 until round 3; the host buffers are shared between frames, the device holds the staged copies)
    v["host_inputs"] = dict(Job(nh, host=True).timed(2, 1), note="SURVEY 8(d) boundary: pageable host buffers -> .drc / .ktx2 bytes in host memory, uploads through pinned double buffers; "
                                                                   "one blocking call per pass and half; enqueued_passes: the same passes through uvol_*_async + uvol_sync (a pass uploads while its predecessor encodes)")
    note("host_inputs enqueued")
    v["host_inputs"]["enqueued_passes"] = Job(nh, host=True, host_enqueued=True).timed(3, 1)["frames_per_s"]
    note("host_inputs pinned")
    trim_geos(tex_too=True)                                    # (the staging path's input buffers on the lanes go back before the uplink's slots are allocated: both do not fit)
    v["host_inputs"]["pinned"] = dict(Job(nh, host=True, pinned=True).timed(2, 1), note="every input array in uvol_host_alloc memory, one blocking call per pass and half: the uploads of the call's groups / parts are queued "
                                                                                         "on the context's copy stream when the call begins (uplink), nothing overlaps the last group's chain")
    note("host_inputs pinned enqueued")
    v["host_inputs"]["pinned_enqueued"] = dict(Job(nh, host=True, pinned=True, host_enqueued=True).timed(5, 1),
                                               note="SURVEY 8(d)'s boundary as specified - inputs resident in PINNED host memory -> bytes in host memory - through the enqueue forms: "
                                                    "the next pass's uploads run on the copy streams beside this pass's kernels")
    bytes_per_frame = 12.0 * V * 2 + 8.0 * sum(len(m["uv"]) for m in meshes_h) / len(meshes_h) + 36.0 * Fc + 4.0 * args.tex_size ** 2
    v["host_inputs"]["link"] = {"bytes_uploaded_per_frame": bytes_per_frame, "link_GBps_measured": 55.0,
                                "frames_per_s_the_link_allows": 55.0e9 / bytes_per_frame,
                                "note": "the C ABI takes OBJ-style separate index streams (36 bytes per face, SURVEY 8(d): 'would add 24 F'), so a frame is ~27.4 MB on the link, not the 22.4 MB of the "
                                        "unified-index formula; 55 GB/s is what this box's link delivers for >= 16 MiB copies out of page-locked memory (profiles/r05_h2d_rate.json, r06_uplink_forms.json)"}
    # (4) decode path (BASELINE configs[4]) on this run's own output: fresh contexts (the encoders' workspaces are released first)
    note("decode")
    drc = [bytes(x) for x in out["drc"][:480]]; ktx = [bytes(x) for x in out["ktx2"][:192]]
    drc = (drc * 4)[:1920]                                      # one call of 1920 frames (~90 MB of decode workspace each)
    for c in geos + texs:
        c.close()
    del geos[:]; del texs[:]
    torch.cuda.empty_cache()
    try:
        d = uvol.Codec(device=local_rank, max_batch=len(drc))
        d.decode_mesh_batch(drc[:8], fetch=False)
        d.decode_mesh_batch(drc, fetch=False)
        torch.cuda.synchronize(); t = time.perf_counter()
        d.decode_mesh_batch(drc, fetch=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        v["decode_mesh"] = {"frames_per_s": len(drc) / dt, "frames": len(drc), "ms": 1000.0 * dt, "note": ".drc -> de-quantised attribute arrays + per-corner indices, results left in HBM"}
        nhd = min(len(drc), 960)                               # ... and with the arrays in caller-owned host memory (kept between calls; 10.5 MB per frame)
        d.decode_mesh_batch(drc[:nhd], views=True)
        t = time.perf_counter(); d.decode_mesh_batch(drc[:nhd], views=True); dt = time.perf_counter() - t
        v["decode_mesh"]["to_host_memory"] = {"frames_per_s": nhd / dt, "frames": nhd}
        d.__dict__.pop("_dec_bufs", None)
        ar_ = uvol.PinnedArena(d.decode_arena_bytes(drc[:nhd]))              # ... and with the caller's arrays in uvol_host_alloc memory: no staging buffers, no host copy
        d.decode_mesh_batch(drc[:nhd], views=True, arena=ar_)
        t = time.perf_counter(); d.decode_mesh_batch(drc[:nhd], views=True, arena=ar_); dt = time.perf_counter() - t
        v["decode_mesh"]["to_pinned_host_memory"] = {"frames_per_s": nhd / dt, "frames": nhd}
        d.__dict__.pop("_dec_bufs_pinned", None); ar_.close()
        size = args.tex_size
        bufs = torch.empty((len(ktx), B, size, size, 4), dtype=torch.uint8, device=dev)
        ptrs = [bufs[s, l].data_ptr() for s in range(len(ktx)) for l in range(B)]
        d.decode_texture_segments_dev(ktx[:2], ptrs[:2 * B], size * size * 4)
        d.decode_texture_segments_dev(ktx, ptrs, size * size * 4)
        torch.cuda.synchronize(); t = time.perf_counter()
        d.decode_texture_segments_dev(ktx, ptrs, size * size * 4)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        v["decode_texture"] = {"frames_per_s": len(ktx) * B / dt, "segments": len(ktx), "ms": 1000.0 * dt, "note": "ETC1S .ktx2 -> RGBA32 layers in HBM"}
        d.close()
    except Exception as e:
        v["decode_error"] = repr(e)
    v.pop("in_progress", None)
    return v


def parity_check(out, meshes_h, tex_h, n_check, only, mesh_order):
    """After the timed region, the oracle as CHECKER only: (1) frames of the step with equal content must have byte-identical
    outputs (frame i and frame i mod ND read equal bytes from different buffers); (2) `n_check` DISTINCT frames spread over the
    step (and texture segment 0) must equal the CPU oracle's bytes for the same input.  A mismatch does not hide the line: it is
    reported in it (`mismatches`) and on stderr."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    O.lib()
    ND = len(meshes_h)
    res = {"geometry_frames_equal_to_oracle": 0, "mismatches": []}
    if only != "tex":
        drc = out["drc"]; n = len(drc)
        same = 0
        for i in range(ND, n):
            a, b = drc[i], drc[i % ND]
            if len(a) == len(b) and np.array_equal(np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)):
                same += 1
            else:
                res["mismatches"].append("drc[%d] != drc[%d] (equal content)" % (i, i % ND))
        res["equal_content_frames_identical"] = same; res["equal_content_frames"] = max(0, n - ND)
        pick = sorted({(k * n) // max(1, n_check) + (k % ND) for k in range(n_check)} & set(range(n))) if n else []
        pick = pick[:n_check]
        for i in pick:
            m = meshes_h[i % ND]
            ref = O.drc_encode(m["pos"], m["idx_pos"], m.get("uv"), m.get("idx_uv"), m.get("nrm"), m.get("idx_nrm"))
            if bytes(drc[i]) == bytes(ref):
                res["geometry_frames_equal_to_oracle"] += 1
            else:
                res["mismatches"].append("drc[%d] != oracle (content %d)" % (i, i % ND))
        res["geometry_frames_checked"] = pick
        res["geometry_distinct_contents_checked"] = len({i % ND for i in pick})
    if only != "geo" and out.get("ktx2"):
        ref = O.ktx2_encode(tex_h)
        ok = bytes(out["ktx2"][0]) == bytes(ref)
        res["texture_segments_equal_to_oracle"] = 1 if ok else 0
        if not ok:
            res["mismatches"].append("ktx2[0] != oracle")
    if res["mismatches"]:
        print("[bench] PARITY MISMATCH: %s" % res["mismatches"][:8], file=sys.stderr, flush=True)
    res["mismatches"] = res["mismatches"][:16]
    return res


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
        dec = tex.decode_texture_segments([bytes(out["ktx2"][0])])[0]
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
    for f in (PMC_FILE, "archive/r02_pmc_traffic.json"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", f)))
            e = t["kernels"].get(group)
            return None if e is None else e["hbm_bytes_per_frame"] * units
        except Exception:
            continue
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
