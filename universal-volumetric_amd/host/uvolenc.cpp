// uvolenc.cpp — host driver with the semantics of `python3 scripts/Encoder.py project-config.json`
// (scripts/Encoder.py:157-373), calling the HIP codec in-process through the C ABI instead of spawning
// draco_encoder / basisu once per frame / per batch.  Output layout and manifest follow what the stock
// player reads (src/Interfaces.ts:75-132, src/V2/player.ts:141-174).
//
//   uvolenc project-config.json [--gpus N] [--device D] [--batch-frames F] [--ingest-threads T] [--targets ktx2[,etc2]] [--uastc] [--device-inflate [--tex-batch-frames F]]
//                               [--force] [--encoder-py-manifest]
//   --targets   texture targets to write (src/Interfaces.ts:19 TextureFileFormat): `ktx2` always; `etc2` adds one raw ETC2 RGB
//               (ETC1-subset) block image per frame, transcoded on the GPU from the ETC1S segments, and a second target in the manifest
//   --uastc     the KTX2 files carry UASTC LDR 4x4 blocks (`basisu -uastc`) instead of ETC1S/BasisLZ
//   --gpus N    rank r of N encodes the segment-aligned block shard_plan() gives it: whole texture segments and the
//               geometry frames with the same indices; one host thread per GPU and stage, no data crosses between GPUs
#include "uvol_host.hpp"
#include "../../include/uvol_codec.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <unistd.h>

using namespace uvolh;

static std::string dirname_of(const std::string &p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? "" : p.substr(0, k); }
static std::string basename_of(const std::string &p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? p : p.substr(k + 1); }
static bool g_timing = false;      // UVOL_TIMING=1: per-batch stage times on stderr
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::string join(const std::string &a, const std::string &b) { if (a.empty()) return b; if (!b.empty() && b[0] == '/') return b; return a + "/" + b; }

// fn(k) for k in [0, n) on up to nthreads host threads (ingest: OBJ text parsing / PNG inflate; egress: file writes)
// (fn also gets the index of the worker that runs it: the workers' scratch buffers outlive the call)
static void parallel_for_w(size_t n, int nthreads, const std::function<void(size_t, size_t)> &fn) {
  const size_t nt = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, nthreads), n));
  if (nt <= 1) { for (size_t k = 0; k < n; k++) fn(k, 0); return; }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; t++) th.emplace_back([&, t] { for (size_t k; (k = next.fetch_add(1)) < n;) fn(k, t); });
  for (auto &t : th) t.join();
}
static void parallel_for(size_t n, int nthreads, const std::function<void(size_t)> &fn) { parallel_for_w(n, nthreads, [&](size_t k, size_t) { fn(k); }); }

int main(int argc, char **argv) {
  // one hardware queue per HIP stream (the runtime's default is four per process; streams that share one run one after the other, and a
  // context holds two per geometry lane plus the texture lanes): read by the runtime when it initialises, i.e. before the first HIP call
  setenv("GPU_MAX_HW_QUEUES", "24", 0);
  if (argc < 2) { std::printf("❌ Invalid number of arguments. Please supply project-config.json as argument\n"); return 1; }
  if (!std::strcmp(argv[1], "create-template")) {
    const std::string t = config_template();
    if (!write_file("project-config-template.json", t.data(), t.size())) return 1;
    std::printf("✅ Written template object to project-config-template.json\n"); return 0;
  }
  const auto t_start = std::chrono::steady_clock::now();
  { const char *e = std::getenv("UVOL_TIMING"); g_timing = e && *e == '1'; }
  int n_gpus = 1, device0 = 0, frames_per_batch = 32, ingest_threads = 0; bool force = false, encpy = false, want_etc2 = false, uastc = false, host_obj = false, host_png = false, dev_inflate = false, pinned_text = false; int tex_batch_frames = 0;
  for (int i = 2; i < argc; i++) {
    if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) n_gpus = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device0 = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--batch-frames") && i + 1 < argc) frames_per_batch = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--ingest-threads") && i + 1 < argc) ingest_threads = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--force")) force = true;
    else if (!std::strcmp(argv[i], "--host-png-unfilter")) host_png = true;       // PNG scanlines un-filtered by the ingest threads (as until round 3) instead of on the GPU
    else if (!std::strcmp(argv[i], "--pinned-text")) pinned_text = true;            // OBJ files read straight into page-locked memory (no staging copy before the upload)
    else if (!std::strcmp(argv[i], "--device-inflate")) dev_inflate = true;        // the PNGs' zlib streams inflated on the GPU too (k_inflate: one wave per image; pays with many images per call, see --tex-batch-frames)
    else if (!std::strcmp(argv[i], "--tex-batch-frames") && i + 1 < argc) tex_batch_frames = std::atoi(argv[++i]);      // images per texture call (default: --batch-frames)
    else if (!std::strcmp(argv[i], "--host-obj-parser")) host_obj = true;          // OBJ text parsed by the ingest threads (as until round 3) instead of on the GPU
    else if (!std::strcmp(argv[i], "--uastc")) uastc = true;
    else if (!std::strcmp(argv[i], "--encoder-py-manifest")) encpy = true;
    else if (!std::strcmp(argv[i], "--targets") && i + 1 < argc) {
      const std::string t = argv[++i]; size_t a = 0;
      while (a <= t.size()) {
        size_t b = t.find(',', a); if (b == std::string::npos) b = t.size();
        const std::string k = t.substr(a, b - a);
        if (k == "etc2") want_etc2 = true; else if (k != "ktx2" && !k.empty()) { std::printf("❌ unknown texture target '%s' (ktx2, etc2)\n", k.c_str()); return 1; }
        a = b + 1;
      }
    }
  }
  if (want_etc2 && uastc) { std::printf("❌ the etc2 target is transcoded from ETC1S segments; it cannot be combined with --uastc\n"); return 1; }
  std::vector<uint8_t> raw; if (!read_file(argv[1], raw)) { std::printf("❌ cannot read %s\n", argv[1]); return 1; }
  Config cfg; std::string err;
  if (!load_config(std::string(raw.begin(), raw.end()), cfg, err)) { std::printf("❌ %s\n", err.c_str()); return 1; }
  if (cfg.compression_level < 0 || cfg.compression_level > 10) { std::printf("❌ DRACO_COMPRESSION_LEVEL must be in [0, 10]\n"); return 1; }
  if (cfg.compression_level == 0) std::printf("💡 DRACO_COMPRESSION_LEVEL 0: sequential connectivity with the difference predictor (what stock draco_encoder selects at this level)\n");
  else if (cfg.compression_level != 7) std::printf("💡 DRACO_COMPRESSION_LEVEL %d: encoded with the compression-level-7 tool set (same bitstream syntax)\n", cfg.compression_level);
  char cwd[4096]; if (!getcwd(cwd, sizeof cwd)) return 1;
  cfg.output_directory = join(cwd, cfg.output_directory);          // scripts/Encoder.py:201
  if (!make_dirs(cfg.output_directory)) { std::printf("❌ cannot create %s\n", cfg.output_directory.c_str()); return 1; }
  if (uvol_device_count() <= 0) { std::printf("❌ no HIP device: uvolenc has no CPU fallback\n"); return 1; }
  n_gpus = std::max(1, std::min(n_gpus, uvol_device_count() - device0));

  uvol_params prm; uvol_params_default(&prm);
  prm.q_position_attr = cfg.q_position; prm.q_texture_attr = cfg.q_texture; prm.q_normal_attr = cfg.q_normal; prm.q_generic_attr = cfg.q_generic;
  prm.draco_compression_level = cfg.compression_level; prm.ktx2_batch_size = cfg.ktx2_batch_size; prm.max_batch = frames_per_batch; prm.uastc = uastc ? 1 : 0;
  // one geometry and one texture context (= HIP stream) per GPU: the two stages of a GPU run side by side
  // (+ one ingest context per GPU when the OBJ text is parsed on the device: contexts are independent, so batch b + 1 is parsed while the
  // geometry context's enqueued call encodes batch b)
  std::vector<uvol_ctx *> ctxs((size_t)n_gpus, nullptr), tctxs((size_t)n_gpus, nullptr), pctxs((size_t)n_gpus, nullptr);
  for (int g = 0; g < n_gpus; g++) if (uvol_ctx_create(device0 + g, &prm, &ctxs[g]) != UVOL_OK || uvol_ctx_create(device0 + g, &prm, &tctxs[g]) != UVOL_OK ||
                                       (!host_obj && !cfg.obj_files_path.empty() && uvol_ctx_create(device0 + g, &prm, &pctxs[g]) != UVOL_OK)) { std::printf("❌ cannot create codec context on GPU %d\n", device0 + g); return 1; }
  // default: the CPUs this process may use (cgroup quota aware) shared by the two stages of every GPU.  Measured under a 16-CPU
  // quota (960 frames): 177 / 182 / 179 / 173 / 134 frames/s with 8 / 12 / 16 / 24 / 64 threads per stage
  const bool threads_given = ingest_threads > 0;
  if (ingest_threads <= 0) ingest_threads = (int)std::max(1u, std::min(48u, effective_cpus() / (2u * (unsigned)n_gpus)));
  // With the OBJ text parsed on the device the geometry stage's host work is reading files (~0.3 ms per frame and thread against 20 ms
  // for the parse), while a PNG still costs ~35 ms of inflate + un-filter: the CPUs go to the texture stage (measured, 960 frames under a
  // 16-CPU quota: 8 + 8 threads 183 frames/s - the texture stage alone bounds it at 120 frames per 525 ms).
  int geo_ingest = ingest_threads, tex_ingest = ingest_threads;
  if (!threads_given && !host_obj && !cfg.obj_files_path.empty() && !cfg.images_path.empty()) { const int tot = 2 * ingest_threads; geo_ingest = std::max(1, tot / 6); tex_ingest = std::max(1, tot - geo_ingest); }

  std::printf("🎯 Dealing with Geomety data\n");
  if (!cfg.abc_file_path.empty()) { std::printf("❌ ABCFilePath needs Blender (bpy); export OBJ files and use OBJFilesPath\n"); return 1; }
  const std::string geo_dir = join(cfg.output_directory, "geometry_draco");
  const std::string tex_dir = join(cfg.output_directory, "texture_ktx2_baseColor_default"), etc_dir = join(cfg.output_directory, "texture_etc2_baseColor_default");
  // every early exit happens before the first worker thread exists (a joinable std::thread destroyed on `return` terminates)
  if (!cfg.obj_files_path.empty() && !make_dirs(geo_dir)) { std::printf("❌ cannot create %s\n", geo_dir.c_str()); return 1; }
  if (!cfg.images_path.empty() && (!make_dirs(tex_dir) || (want_etc2 && !make_dirs(etc_dir)))) { std::printf("❌ cannot create %s\n", tex_dir.c_str()); return 1; }
  if (want_etc2 && cfg.images_path.empty()) { std::printf("❌ --targets etc2 needs ImagesPath (the raw target is transcoded while the segments are encoded)\n"); return 1; }
  int pad = 5;
  std::atomic<int> geo_failed{-1};
  std::vector<std::thread> geo_threads;
  std::vector<std::string> obj_files; std::string obj_dir;
  const int B = cfg.ktx2_batch_size;
  if (!cfg.obj_files_path.empty()) {
    std::printf("🚧 Obtained OBJ files path\n");
    obj_dir = dirname_of(cfg.obj_files_path); const std::string pat = basename_of(cfg.obj_files_path);
    { int h = (int)std::count(pat.begin(), pat.end(), '#'); if (h > 0) pad = h; }
    for (auto &f : list_dir(obj_dir)) if (match_pattern_lenient(pat, f)) obj_files.push_back(f);
    // Segment-aligned blocks of frames per GPU (shard_plan): rank g gets the frames of ITS texture segments, so a
    // GPU's geometry and texture outputs cover the same time span; each GPU encodes batches of frames_per_batch frames.  The OBJ text
    // of batch b+1 is parsed by the ingest threads while the GPU encodes batch b, .drc files are written in parallel.
    // Two batch objects per GPU are recycled (the one being loaded, the one being encoded): their meshes, the ingest workers' scratch
    // buffers and the output buffers keep their capacity, so after the first two batches the stage allocates nothing.
    struct GeoBatch { size_t b0 = 0, nb = 0; std::vector<ObjMesh> ms; std::string err; int bad = -1; std::vector<std::unique_ptr<uint8_t[]>> outs; std::vector<size_t> ocap;
                      std::vector<std::vector<uint8_t>> text;              // device parser: the files as they are
                      uint8_t *pin = nullptr; size_t pin_cap = 0; std::vector<size_t> toff, tlen;      // --pinned-text: ... read into ONE page-locked slab (uvol_host_alloc): the upload is DMA from where the text lies
                      ~GeoBatch() { if (pin) uvol_host_free(pin); }
                      std::vector<uvol_mesh> um; std::vector<uint8_t *> op; std::vector<size_t> caps, lens; std::vector<int> st, pst; };
    for (int g = 0; g < n_gpus; g++) geo_threads.emplace_back([&, g] {
      const std::vector<std::string> &files = obj_files;
      const ShardPlan sp = shard_plan((long)files.size(), B, n_gpus, g);
      const size_t lo = (size_t)sp.first_frame, hi = lo + (size_t)sp.n_frames;
      // three batch objects in turn: the one being encoded, the next one (loaded, then prepared while the GPU works) and the one being loaded
      std::shared_ptr<GeoBatch> pool[3] = { std::make_shared<GeoBatch>(), std::make_shared<GeoBatch>(), std::make_shared<GeoBatch>() }; size_t n_loads = 0;
      std::vector<IngestScratch> scratch((size_t)std::max(1, geo_ingest)); IngestScratch fb_scratch;
      // the FIRST batch of a stage is a quarter of the others: the pipeline (read -> parse -> encode -> write) starts after a quarter of the
      // time a full batch of cold files takes to read (0.6 s of a 4 s run at 960 frames)
      auto blen = [&](size_t b0) -> size_t { return b0 == lo ? (size_t)std::max(1, std::min(frames_per_batch, std::max(16, frames_per_batch / 4))) : (size_t)frames_per_batch; };
      auto load = [&](size_t b0, size_t slot) {
        std::shared_ptr<GeoBatch> Bt = pool[slot]; Bt->b0 = b0; Bt->nb = b0 < hi ? std::min(hi - b0, blen(b0)) : 0; Bt->bad = -1; Bt->err.clear();
        if (Bt->ms.size() < Bt->nb) Bt->ms.resize(Bt->nb);
        std::mutex mu;
        const double tl0 = now_ms();
        if (!host_obj) {                                     // the GPU parses: the ingest threads only read the files
          if (pinned_text) {
            Bt->toff.assign(Bt->nb, 0); Bt->tlen.assign(Bt->nb, 0); size_t tot = 0;
            for (size_t k = 0; k < Bt->nb; k++) { const long n = file_size(join(obj_dir, files[b0 + k])); if (n <= 0) { if (Bt->bad < 0) { Bt->bad = (int)k; Bt->err = "cannot read " + files[b0 + k]; } continue; } Bt->toff[k] = tot; Bt->tlen[k] = (size_t)n; tot += ((size_t)n + 4095) & ~(size_t)4095; }
            if (tot > Bt->pin_cap) { if (Bt->pin) uvol_host_free(Bt->pin); Bt->pin_cap = tot + tot / 8; Bt->pin = (uint8_t *)uvol_host_alloc(Bt->pin_cap); if (!Bt->pin) { Bt->pin_cap = 0; Bt->bad = 0; Bt->err = "page-locked text buffer: allocation failed"; } }
            if (Bt->pin) parallel_for_w(Bt->nb, geo_ingest, [&](size_t k, size_t) { if (Bt->tlen[k] && !read_file_into(join(obj_dir, files[b0 + k]), Bt->pin + Bt->toff[k], Bt->tlen[k])) { std::lock_guard<std::mutex> l(mu); if (Bt->bad < 0 || (int)k < Bt->bad) { Bt->bad = (int)k; Bt->err = "cannot read " + files[b0 + k]; } } });
            if (g_timing && Bt->nb) std::fprintf(stderr, "[uvolenc-timing] geo load  b0=%zu n=%zu %.0f ms (page-locked slab of %.0f MB)\n", b0, Bt->nb, now_ms() - tl0, Bt->pin_cap / 1e6);
            return Bt;
          }
          if (Bt->text.size() < Bt->nb) Bt->text.resize(Bt->nb);
          parallel_for_w(Bt->nb, geo_ingest, [&](size_t k, size_t) { if (!read_file(join(obj_dir, files[b0 + k]), Bt->text[k]) || Bt->text[k].empty()) { std::lock_guard<std::mutex> l(mu); if (Bt->bad < 0 || (int)k < Bt->bad) { Bt->bad = (int)k; Bt->err = "cannot read " + files[b0 + k]; } } });
        } else
        parallel_for_w(Bt->nb, geo_ingest, [&](size_t k, size_t w) { std::string e; if (!read_obj(join(obj_dir, files[b0 + k]), Bt->ms[k], e, &scratch[w])) { std::lock_guard<std::mutex> l(mu); if (Bt->bad < 0 || (int)k < Bt->bad) { Bt->bad = (int)k; Bt->err = e; } } });
        if (g_timing && Bt->nb) std::fprintf(stderr, "[uvolenc-timing] geo load  b0=%zu n=%zu %.0f ms\n", b0, Bt->nb, now_ms() - tl0);
        return Bt;
      };
      std::future<std::shared_ptr<GeoBatch>> nextb = std::async(std::launch::async, load, lo, (n_loads++) % 3);
      // The GPU call is ENQUEUED (uvol_encode_mesh_batch_async): while the device encodes batch b this thread writes the .drc files of
      // batch b - 1 (and the ingest threads parse batch b + 1), then uvol_sync completes batch b.  A batch object's output buffers are
      // next written by the encode of batch b + 2, after its files are on disk.
      struct Written { std::vector<std::unique_ptr<uint8_t[]>> *outs = nullptr; std::vector<size_t> lens; size_t b0 = 0, nb = 0; } prev;
      auto write_prev = [&]() {
        if (!prev.outs || !prev.nb) return;
        std::atomic<int> wbad{-1};
        parallel_for(prev.nb, geo_ingest, [&](size_t k) { char name[64]; std::snprintf(name, sizeof name, "%0*zu.drc", pad, prev.b0 + k); if (!write_file(join(geo_dir, name), (*prev.outs)[k].get(), prev.lens[k])) wbad = (int)(prev.b0 + k); });
        prev.nb = 0;
        if (wbad >= 0 && geo_failed < 0) geo_failed = wbad.load();                        // (the first error stands)
      };
      // Prepares a loaded batch for the encode call: (device parser) uploads + parses the OBJ text on the ingest context into slot `slot` -
      // files the device parser hands back (UVOL_E_UNSUPPORTED: a number it cannot decide exactly) are parsed by read_obj and encoded from
      // host arrays afterwards -, fills the mesh / output arrays of the batch object.  false: a frame failed (message printed).
      auto prepare = [&](GeoBatch &Bt, int slot) -> bool {
        const size_t nb = Bt.nb, b0 = Bt.b0;
        if (Bt.bad >= 0) { std::printf("Failed to compress %s\n%s\n", files[b0 + (size_t)Bt.bad].c_str(), Bt.err.c_str()); geo_failed = (int)(b0 + (size_t)Bt.bad); return false; }
        Bt.um.assign(nb, uvol_mesh{}); Bt.op.assign(nb, nullptr); Bt.caps.assign(nb, 0); Bt.lens.assign(nb, 0); Bt.st.assign(nb, 0); Bt.pst.assign(nb, 0);
        if (Bt.outs.size() < nb) { Bt.outs.resize(nb); Bt.ocap.resize(nb, 0); }
        if (!host_obj) {
          std::vector<const uint8_t *> tp(nb); std::vector<size_t> tl(nb);
          for (size_t k = 0; k < nb; k++) { if (pinned_text) { tp[k] = Bt.pin + Bt.toff[k]; tl[k] = Bt.tlen[k]; } else { tp[k] = Bt.text[k].data(); tl[k] = Bt.text[k].size(); } }
          const int rc = uvol_parse_obj_batch_dev(pctxs[g], tp.data(), tl.data(), (int)nb, slot, Bt.um.data(), Bt.pst.data());
          if (rc != UVOL_OK) { std::printf("Failed to compress %s\n%s\n", files[b0].c_str(), uvol_last_error(pctxs[g])); geo_failed = (int)b0; return false; }
          if (Bt.ms.size() < nb) Bt.ms.resize(nb);
          for (size_t k = 0; k < nb; k++) {
            if (Bt.pst[k] == UVOL_OK) continue;
            std::string e;                                                      // the host parser decides (and words the error)
            if (!read_obj(join(obj_dir, files[b0 + k]), Bt.ms[k], e, &fb_scratch)) { std::printf("Failed to compress %s\n%s\n", files[b0 + k].c_str(), e.c_str()); geo_failed = (int)(b0 + k); return false; }
            Bt.pst[k] = UVOL_E_UNSUPPORTED;                                     // parsed on the host: encoded from host arrays (below)
          }
        }
        for (size_t k = 0; k < nb; k++) {
          uvol_mesh &m = Bt.um[k];
          if (host_obj || Bt.pst[k] != UVOL_OK) {
            const ObjMesh &o = Bt.ms[k]; std::memset(&m, 0, sizeof m);
            m.pos = o.pos.data(); m.n_pos = (uint32_t)o.pos.size() / 3; m.idx_pos = o.idx_pos.data(); m.n_faces = (uint32_t)o.idx_pos.size() / 3;
            if (!o.uv.empty()) { m.uv = o.uv.data(); m.n_uv = (uint32_t)o.uv.size() / 2; m.idx_uv = o.idx_uv.data(); }
            if (!o.nrm.empty()) { m.nrm = o.nrm.data(); m.n_nrm = (uint32_t)o.nrm.size() / 3; m.idx_nrm = o.idx_nrm.data(); }
          }
          Bt.caps[k] = uvol_mesh_bound(&m); if (Bt.ocap[k] < Bt.caps[k]) { Bt.outs[k].reset(new uint8_t[Bt.caps[k]]); Bt.ocap[k] = Bt.caps[k]; } Bt.op[k] = Bt.outs[k].get();   // (not zero-filled: the bound is a worst case)
        }
        return true;
      };
      // enqueue the batch: device-parsed frames through the _dev entry point (contiguous runs), host-parsed ones through the host entry point
      auto enqueue = [&](GeoBatch &Bt) -> int {
        const size_t nb = Bt.nb;
        for (size_t a = 0; a < nb;) {
          const bool dev = !host_obj && Bt.pst[a] == UVOL_OK; size_t b = a + 1;
          while (b < nb && (!host_obj && Bt.pst[b] == UVOL_OK) == dev) b++;
          const int rc = (dev ? uvol_encode_mesh_batch_dev_async : uvol_encode_mesh_batch_async)(ctxs[g], Bt.um.data() + a, (int)(b - a), Bt.op.data() + a, Bt.caps.data() + a, Bt.lens.data() + a, Bt.st.data() + a);
          if (rc != UVOL_OK) return rc;
          a = b;
        }
        return UVOL_OK;
      };
      std::shared_ptr<GeoBatch> cur = lo < hi ? nextb.get() : nullptr; size_t n_prep = 0;
      if (cur) { nextb = std::async(std::launch::async, load, lo + blen(lo), (n_loads++) % 3); if (!prepare(*cur, (int)((n_prep++) & 1))) cur = nullptr; }
      for (size_t b0 = lo; cur && b0 < hi && geo_failed < 0; b0 += blen(b0)) {
        GeoBatch &Bt = *cur; const size_t nb = Bt.nb;
        const double te0 = now_ms();
        int rc = enqueue(Bt);
        write_prev();                                                                  // the previous batch's files, while the GPU works
        const double te1 = now_ms();
        // ... and the NEXT batch: its files are read (ingest threads), its text uploaded and parsed (ingest context) while this one encodes
        std::shared_ptr<GeoBatch> nxt; bool nxt_ok = true;
        if (b0 + blen(b0) < hi) {
          nxt = nextb.get();
          nextb = std::async(std::launch::async, load, b0 + blen(b0) + blen(b0 + blen(b0)), (n_loads++) % 3);      // (the object of batch b - 1: its files are written, its meshes done with)
          if (rc == UVOL_OK) nxt_ok = prepare(*nxt, (int)((n_prep++) & 1));
        }
        const double te2 = now_ms();
        if (rc == UVOL_OK) rc = uvol_sync(ctxs[g]);
        if (rc != UVOL_OK) { std::printf("Failed to compress %s\n%s\n", files[b0].c_str(), uvol_last_error(ctxs[g])); if (geo_failed < 0) geo_failed = (int)b0; break; }
        for (size_t k = 0; k < nb; k++) if (Bt.st[k] != UVOL_OK) { std::printf("Failed to compress %s\n", files[b0 + k].c_str()); if (geo_failed < 0 || (int)(b0 + k) < geo_failed) geo_failed = (int)(b0 + k); break; }   // scripts/Encoder.py:263-266
        if (geo_failed >= 0 && !nxt_ok) break;
        if (geo_failed >= 0 && (size_t)geo_failed >= b0 && (size_t)geo_failed < b0 + nb) break;
        prev.outs = &Bt.outs; prev.lens = Bt.lens; prev.b0 = b0; prev.nb = nb;
        if (g_timing) std::fprintf(stderr, "[uvolenc-timing] geo batch b0=%zu: enqueue + write of the previous batch %.0f ms, load + prepare of the next (GPU busy) %.0f, waited for the GPU %.0f\n", b0, te1 - te0, te2 - te1, now_ms() - te2);
        if (!nxt_ok) break;
        cur = nxt;
      }
      write_prev();          // also after a failure: the batch before the failing one was encoded and is written, as the reference's frame-at-a-time loop would have left it (scripts/Encoder.py:256-267)
      if (nextb.valid()) nextb.wait();
    });
  }

  std::printf("🎯 Dealing with Texture data\n");
  uint32_t tex_w = 0, tex_h = 0;
  std::atomic<int> tex_failed{-1};
  std::atomic<long> etc2_frames{0};
  std::vector<std::thread> tex_threads;
  std::vector<int> starts; std::string cpat;
  if (!cfg.images_path.empty()) {
    std::printf("🚧 Obtained Images path.\n");
    cpat = convert_pounds_to_c_style(cfg.images_path);     // scripts/Encoder.py:274
    for (int i = cfg.ktx2_first_file; i < cfg.ktx2_file_count; i += cfg.ktx2_batch_size) starts.push_back(i);   // :282-287
    // Segments of KTX2_BATCH_SIZE images are independent.  Full segments go to the GPU `segs_per_call` at a time
    // through the batched entry point (one launch per stage for all of them); PNGs of the next call are inflated by the
    // ingest threads meanwhile.  A short last segment (fewer layers) is encoded on its own.
    const int segs_per_call = std::max(1, (tex_batch_frames > 0 ? tex_batch_frames : frames_per_batch) / std::max(1, cfg.ktx2_batch_size));
    struct TexBatch { size_t s0 = 0, ns = 0; std::vector<std::vector<Image>> imgs; std::vector<std::vector<Image>> spare; std::string err; int bad = -1;
                      bool dev = false; int slot = 0; std::vector<std::vector<PngRaw>> raws;        // dev: the images are INFLATED scanlines, un-filtered on the GPU
                      std::vector<std::vector<const uint8_t *>> dptr; bool issued = false; };   // ... their RGBA layers in HBM once the un-filter call is queued
    for (int g = 0; g < n_gpus; g++) tex_threads.emplace_back([&, g] {
      const size_t lo = starts.size() * (size_t)g / n_gpus, hi = starts.size() * (size_t)(g + 1) / n_gpus;       // = shard_plan's segment block
      // three batch objects in turn: the one being encoded, the next one (loaded; its un-filter queued beside this encode), the one being loaded
      std::shared_ptr<TexBatch> pool[3] = { std::make_shared<TexBatch>(), std::make_shared<TexBatch>(), std::make_shared<TexBatch>() }; size_t n_loads = 0;
      std::vector<IngestScratch> scratch((size_t)std::max(1, tex_ingest));
      // (a short first call, as in the geometry stage; not with the device inflate, whose kernel takes the same time for 1 or 1000 images)
      auto slen = [&](size_t s0) -> size_t { return s0 == lo && !dev_inflate ? (size_t)std::max(1, std::min(segs_per_call, std::max(4, segs_per_call / 4))) : (size_t)segs_per_call; };
      auto load = [&](size_t s0, size_t slot) {
        std::shared_ptr<TexBatch> T = pool[slot]; T->s0 = s0; T->ns = s0 < hi ? std::min(hi - s0, slen(s0)) : 0; T->bad = -1; T->err.clear();
        // recycled Image objects keep their 16.8 MB pixel buffers (a short last segment shrinks imgs[s]; the layers come back from `spare`)
        for (auto &v : T->imgs) for (auto &im : v) { T->spare.emplace_back(); T->spare.back().push_back(std::move(im)); }
        T->imgs.clear(); T->imgs.resize(T->ns);
        for (auto &v : T->imgs) { v.resize((size_t)B); for (auto &im : v) if (!T->spare.empty()) { im = std::move(T->spare.back()[0]); T->spare.pop_back(); } }
        const double tl0 = now_ms();
        std::mutex mu; std::vector<std::vector<uint8_t>> present(T->ns, std::vector<uint8_t>((size_t)B, 0));
        // device un-filter: the ingest threads parse the chunks and inflate, nothing else; a batch with an image the device path does not take
        // (16-bit, palette, grey, sizes that differ) is decoded by read_png as before
        T->dev = false;
        if (!host_png) {
          if (T->raws.size() < T->ns) T->raws.resize(T->ns);
          for (size_t s = 0; s < T->ns; s++) if (T->raws[s].size() < (size_t)B) T->raws[s].resize((size_t)B);
          std::atomic<int> odd{0};
          parallel_for_w(T->ns * (size_t)B, tex_ingest, [&](size_t j, size_t wk) {
            const size_t s = j / (size_t)B; const int k = (int)(j % (size_t)B);
            char path[4096]; std::snprintf(path, sizeof path, cpat.c_str(), (unsigned)(starts[s0 + s] + k));
            std::string e;
            const int r = read_png_raw(path, T->raws[s][(size_t)k], e, &scratch[wk], dev_inflate);
            if (r == 1) present[s][(size_t)k] = 1;
            else if (r == 0) odd = 1;
            else if (k == 0 || starts[s0 + s] + k < cfg.ktx2_file_count) { std::lock_guard<std::mutex> l(mu); if (T->bad < 0 || (int)s < T->bad) { T->bad = (int)s; T->err = e; } }
          });
          bool ok = !odd && T->bad < 0 && T->ns > 0 && present[0][0];
          if (ok) { const PngRaw &r0 = T->raws[0][0]; for (size_t s = 0; s < T->ns && ok; s++) for (size_t k = 0; k < (size_t)B && ok; k++) if (present[s][k]) { const PngRaw &r = T->raws[s][k]; ok = r.w == r0.w && r.h == r0.h && r.ch == r0.ch; } }
          if (ok) {
            T->dev = true;
            for (size_t s = 0; s < T->ns; s++) { size_t n = 0; while (n < (size_t)B && present[s][n]) n++; T->imgs[s].resize(n); for (size_t k = 0; k < n; k++) { T->imgs[s][k].w = T->raws[s][k].w; T->imgs[s][k].h = T->raws[s][k].h; } }
            if (g_timing && T->ns) std::fprintf(stderr, "[uvolenc-timing] tex load  s0=%zu n=%zu segments %.0f ms (%s)\n", s0, T->ns, now_ms() - tl0, dev_inflate ? "files read, chunks parsed" : "inflate only");
            return T;
          }
          if (T->bad >= 0) return T;                         // a file is missing: reported as before
          for (auto &v : present) std::fill(v.begin(), v.end(), 0);
        }
        parallel_for_w(T->ns * (size_t)B, tex_ingest, [&](size_t j, size_t wk) {
          const size_t s = j / (size_t)B; const int k = (int)(j % (size_t)B);
          char path[4096]; std::snprintf(path, sizeof path, cpat.c_str(), (unsigned)(starts[s0 + s] + k));
          std::string e;
          if (read_png(path, T->imgs[s][(size_t)k], e, &scratch[wk])) present[s][(size_t)k] = 1;
          else if (k == 0 || starts[s0 + s] + k < cfg.ktx2_file_count) { std::lock_guard<std::mutex> l(mu); if (T->bad < 0 || (int)s < T->bad) { T->bad = (int)s; T->err = e; } }
        });
        for (size_t s = 0; s < T->ns; s++) { size_t n = 0; while (n < (size_t)B && present[s][n]) n++; T->imgs[s].resize(n); }   // layers of a short last segment
        if (g_timing && T->ns) std::fprintf(stderr, "[uvolenc-timing] tex load  s0=%zu n=%zu segments %.0f ms\n", s0, T->ns, now_ms() - tl0);
        return T;
      };
      auto fail = [&](int first, const char *why) { std::printf("Failed to compress images with indices: [%d, %d]\n%s\n", first, first + B, why); tex_failed = first; };   // :293-298
      std::future<std::shared_ptr<TexBatch>> nextb = std::async(std::launch::async, load, lo, (n_loads++) % 3);
      // device un-filter of a loaded batch: every image in one call, queued on the texture context's ingest stream (returns once the
      // scanlines are staged); the encode calls order themselves behind it
      size_t n_issue = 0;
      auto issue = [&](TexBatch &Tb) -> bool {
        Tb.issued = true; Tb.dptr.assign(Tb.ns, {});
        if (!Tb.dev || Tb.bad >= 0) return true;
        std::vector<const uint8_t *> rp; for (size_t s = 0; s < Tb.ns; s++) for (size_t k = 0; k < Tb.imgs[s].size(); k++) rp.push_back(Tb.raws[s][k].raw.data());
        std::vector<const uint8_t *> dp(rp.size(), nullptr);
        Tb.slot = (int)((n_issue++) & 1);
        int rcu;
        if (dev_inflate) { std::vector<size_t> zl; for (size_t s = 0; s < Tb.ns; s++) for (size_t k = 0; k < Tb.imgs[s].size(); k++) zl.push_back(Tb.raws[s][k].raw.size());
          rcu = uvol_inflate_png_batch_dev(tctxs[g], rp.data(), zl.data(), (int)rp.size(), Tb.raws[0][0].w, Tb.raws[0][0].h, Tb.raws[0][0].ch, Tb.slot, dp.data()); }
        else rcu = uvol_unfilter_png_batch_dev(tctxs[g], rp.data(), (int)rp.size(), Tb.raws[0][0].w, Tb.raws[0][0].h, Tb.raws[0][0].ch, Tb.slot, dp.data());
        if (rcu != UVOL_OK) { fail(starts[Tb.s0], uvol_last_error(tctxs[g])); return false; }
        size_t q = 0; for (size_t s = 0; s < Tb.ns; s++) for (size_t k = 0; k < Tb.imgs[s].size(); k++) Tb.dptr[s].push_back(dp[q++]);
        return true;
      };
      std::shared_ptr<TexBatch> cur = lo < hi ? nextb.get() : nullptr;
      if (cur) { nextb = std::async(std::launch::async, load, lo + slen(lo), (n_loads++) % 3); if (!issue(*cur)) cur = nullptr; }
      for (size_t s0 = lo; cur && s0 < hi && tex_failed < 0; s0 += slen(s0)) {
        const double tw0 = now_ms();
        std::shared_ptr<TexBatch> T = cur;
        // the NEXT batch: loaded (PNG chunks parsed + inflated) by the ingest threads by now or soon; its un-filter kernel runs beside this batch's encode
        std::shared_ptr<TexBatch> nxt;
        if (s0 + slen(s0) < hi) {
          nxt = nextb.get();
          nextb = std::async(std::launch::async, load, s0 + slen(s0) + slen(s0 + slen(s0)), (n_loads++) % 3);
          if (!issue(*nxt)) break;
        }
        const double tw1 = now_ms();
        cur = nxt;
        if (T->bad >= 0) { fail(starts[s0 + (size_t)T->bad], T->err.c_str()); break; }
        const uint32_t w = T->imgs[0][0].w, h = T->imgs[0][0].h; bool same = true;
        for (auto &seg : T->imgs) for (auto &im : seg) if (im.w != w || im.h != h) same = false;
        if (!same) { fail(starts[s0], "image sizes differ"); break; }
        tex_w = w; tex_h = h;
        if (T->dev && dev_inflate) {                   // a PNG whose zlib stream the device found corrupt: its segment fails as `basisu` would on that file (scripts/Encoder.py:293-298)
          size_t ni = 0; for (auto &seg : T->imgs) ni += seg.size();
          std::vector<int> pst(ni, UVOL_OK); int bad_seg = -1;
          if (uvol_png_status(tctxs[g], T->slot, pst.data(), (int)ni) != UVOL_OK) { fail(starts[s0], uvol_last_error(tctxs[g])); break; }
          size_t q = 0; for (size_t s = 0; s < T->ns && bad_seg < 0; s++) for (size_t k = 0; k < T->imgs[s].size(); k++) if (pst[q++] != UVOL_OK) { bad_seg = (int)s; break; }
          if (bad_seg >= 0) { fail(starts[s0 + (size_t)bad_seg], "zlib inflate failed"); break; }
        }
        std::vector<std::unique_ptr<uint8_t[]>> outs(T->ns); std::vector<size_t> lens(T->ns, 0);
        // full segments: one batched call; segments with fewer layers: one call each
        std::vector<size_t> full; for (size_t s = 0; s < T->ns; s++) if ((int)T->imgs[s].size() == B) full.push_back(s);
        const std::vector<std::vector<const uint8_t *>> &dptr = T->dptr;
        if (!full.empty()) {
          std::vector<const uint8_t *> ptrs; std::vector<uint8_t *> op; std::vector<size_t> caps, ln(full.size(), 0);
          for (size_t s : full) { for (size_t k = 0; k < T->imgs[s].size(); k++) ptrs.push_back(T->dev ? dptr[s][k] : T->imgs[s][k].rgba.data()); const size_t cap = uvol_texture_bound(w, h, B); outs[s].reset(new uint8_t[cap]); op.push_back(outs[s].get()); caps.push_back(cap); }
          if ((T->dev ? uvol_encode_texture_segments_dev : uvol_encode_texture_segments)(tctxs[g], ptrs.data(), (int)full.size(), B, w, h, op.data(), caps.data(), ln.data()) != UVOL_OK) { fail(starts[s0 + full[0]], uvol_last_error(tctxs[g])); break; }
          for (size_t q = 0; q < full.size(); q++) lens[full[q]] = ln[q];
        }
        for (size_t s = 0; s < T->ns && tex_failed < 0; s++) if ((int)T->imgs[s].size() != B) {
          std::vector<const uint8_t *> ptrs; for (size_t k = 0; k < T->imgs[s].size(); k++) ptrs.push_back(T->dev ? dptr[s][k] : T->imgs[s][k].rgba.data());
          const size_t cap = uvol_texture_bound(w, h, (int)ptrs.size()); outs[s].reset(new uint8_t[cap]);
          if ((T->dev ? uvol_encode_texture_segment_dev : uvol_encode_texture_segment)(tctxs[g], ptrs.data(), (int)ptrs.size(), w, h, outs[s].get(), cap, &lens[s]) != UVOL_OK) fail(starts[s0 + s], uvol_last_error(tctxs[g]));
        }
        if (g_timing) std::fprintf(stderr, "[uvolenc-timing] tex batch s0=%zu: next batch loaded + its un-filter queued %.0f ms, encode (+prepare) %.0f\n", s0, tw1 - tw0, now_ms() - tw1);
        if (tex_failed >= 0) break;
        std::atomic<int> wbad{-1};
        parallel_for(T->ns, tex_ingest, [&](size_t s) { char name[64]; std::snprintf(name, sizeof name, "%0*d.ktx2", pad, (starts[s0 + s] - cfg.ktx2_first_file) / B); if (!write_file(join(tex_dir, name), outs[s].get(), lens[s])) wbad = starts[s0 + s]; });
        if (wbad >= 0) { tex_failed = wbad.load(); break; }
        // raw `etc2` target (src/V2/player.ts:338-356): every layer of the segments just written, transcoded on the GPU to ETC1 blocks
        // (valid ETC2 RGB), one .etc2 file per FRAME (the player builds one CompressedTexture per file)
        if (want_etc2) {
          const size_t nbk = (size_t)((w + 3) / 4) * ((h + 3) / 4) * 8;
          for (size_t s = 0; s < T->ns && tex_failed < 0; s++) {
            const size_t nl = T->imgs[s].size();
            std::vector<std::unique_ptr<uint8_t[]>> blk(nl); std::vector<uint8_t *> bp(nl);
            for (size_t l = 0; l < nl; l++) { blk[l].reset(new uint8_t[nbk]); bp[l] = blk[l].get(); }
            const uint8_t *fp = outs[s].get(); const size_t fl = lens[s];
            if (uvol_transcode_texture_segments_etc1(tctxs[g], &fp, &fl, 1, bp.data(), nbk, 0) != UVOL_OK) { fail(starts[s0 + s], uvol_last_error(tctxs[g])); break; }
            for (size_t l = 0; l < nl; l++) { char name[64]; std::snprintf(name, sizeof name, "%0*d.etc2", pad, starts[s0 + s] - cfg.ktx2_first_file + (int)l); if (!write_file(join(etc_dir, name), bp[l], nbk)) { tex_failed = starts[s0 + s]; break; } }
            etc2_frames += (long)nl;
          }
        }
      }
      if (nextb.valid()) nextb.wait();
    });
  }
  // both stages were started above and run concurrently (geometry and texture contexts of each GPU); join and report in
  // the order the reference prints (scripts/Encoder.py:244-302)
  for (auto &t : geo_threads) t.join();
  for (auto &t : tex_threads) t.join();
  for (auto *c : ctxs) uvol_ctx_destroy(c);
  for (auto *c : tctxs) uvol_ctx_destroy(c);
  for (auto *c : pctxs) if (c) uvol_ctx_destroy(c);
  if (geo_failed >= 0 || tex_failed >= 0) return 1;
  const double t_encode = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  if (!cfg.obj_files_path.empty()) cfg.draco_files_path = join(geo_dir, std::string((size_t)pad, '#') + ".drc");
  if (!cfg.draco_files_path.empty()) std::printf("✅ Obtained DRACO files\n");
  if (!cfg.images_path.empty()) cfg.ktx2_files_path = join(tex_dir, std::string((size_t)pad, '#') + ".ktx2");
  if (!cfg.ktx2_files_path.empty()) std::printf("✅ Obtained KTX2 files\n");

  FrameCounts fc;
  if (!check_total_frames(cfg.draco_files_path, cfg.ktx2_files_path, cfg.ktx2_batch_size, cfg.geometry_frame_rate, cfg.texture_frame_rate, fc, err)) { std::printf("❌ %s\n", err.c_str()); return 1; }
  std::printf("Geometry frame count: %ld\nTexture frame count (not segments): %ld\n", fc.geometry_frames, fc.texture_frames);
  if (!fc.compatible) {
    std::printf("❌ Number of Geometry frames and Texture frames are not compatible with the given frame rates\n");
    if (!force) { std::printf("(re-run with --force to proceed anyway)\n"); return 1; }        // the reference prompts y/n (:141-146)
  } else std::printf("✅ Frames and frame rates are compatible\n");
  // audio duration against the two durations (scripts/Encoder.py:331-347); the reference asks y/n on a mismatch, this driver needs --force
  if (!cfg.audio_url.empty()) {
    double sec = 0; std::string aerr;
    if (!audio_duration(join(cwd, cfg.audio_url), sec, aerr)) std::printf("💡 Audio duration not checked (%s)\n", aerr.c_str());
    else if (fc.geometry_duration == sec && fc.texture_duration == sec) std::printf("✅ Audio duration matches with the frame count and frame rates\n");
    else {
      std::printf("❌ Audio duration doesn't match with the frame count and frame rates\nUVOL durations (without audio):  {'geometry': %.17g, 'texture': %.17g}\nAudio duration: %.17g\n", fc.geometry_duration, fc.texture_duration, sec);
      if (!force) { std::printf("(re-run with --force to proceed anyway)\n"); return 1; }
    }
  } else std::printf("💡 Audio file not supplied, Skipping duration check...\n");
  if (!tex_w) { std::vector<uint8_t> k; std::string d = dirname_of(cfg.ktx2_files_path); for (auto &f : list_dir(d)) if (match_pattern_lenient(basename_of(cfg.ktx2_files_path), f)) { if (read_file(join(d, f), k) && k.size() >= 28) { std::memcpy(&tex_w, &k[20], 4); std::memcpy(&tex_h, &k[24], 4); } break; } }
  const std::string man = json_dump(manifest_player(cfg, fc.geometry_frames, fc.texture_segments, tex_w, tex_h, pad, want_etc2 ? etc2_frames.load() : 0));
  const std::string mpath = join(cfg.output_directory, "uvol.json");
  if (!write_file(mpath, man.data(), man.size())) return 1;
  if (encpy) { const std::string m2 = json_dump(manifest_encoder_py(cfg, fc.geometry_frames, fc.texture_segments, "geometry_draco/" + std::string((size_t)pad, '#') + ".drc", "texture_ktx2_baseColor_default/" + std::string((size_t)pad, '#') + ".ktx2"));
    write_file(join(cfg.output_directory, "uvol.encoderpy.json"), m2.data(), m2.size()); }
  std::printf("✅ Written Manifest file: %s.\n", mpath.c_str());
  if (std::fmod(cfg.geometry_frame_rate, cfg.texture_frame_rate) != 0 && std::fmod(cfg.texture_frame_rate, cfg.geometry_frame_rate) != 0)      // scripts/Encoder.py:368-373
    std::printf("⚠️ Warning: Frame rates are not factors of one another. Ambiguities may arise when calulating appropriate texture for geometry frames.\n");
  // machine-readable timing of the encode phase (files read -> all .drc / .ktx2 written), for the end-to-end figure of SURVEY §8d
  std::printf("[uvolenc] frames %ld, encode phase %.3f s, %.1f frames/s, %d GPU(s), %d + %d ingest threads (geometry + texture stage)\n", fc.geometry_frames, t_encode, t_encode > 0 ? (double)fc.geometry_frames / t_encode : 0.0, n_gpus, geo_ingest, tex_ingest);
  return 0;
}
