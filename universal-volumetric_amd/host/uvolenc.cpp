// uvolenc.cpp — host driver with the semantics of `python3 scripts/Encoder.py project-config.json`
// (scripts/Encoder.py:157-373), calling the HIP codec in-process through the C ABI instead of spawning
// draco_encoder / basisu once per frame / per batch.  Output layout and manifest follow what the stock
// player reads (src/Interfaces.ts:75-132, src/V2/player.ts:141-174).
#include "uvol_host.hpp"
#include "../../include/uvol_codec.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>

using namespace uvolh;

static std::string dirname_of(const std::string &p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? "" : p.substr(0, k); }
static std::string basename_of(const std::string &p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? p : p.substr(k + 1); }
static std::string join(const std::string &a, const std::string &b) { if (a.empty()) return b; if (!b.empty() && b[0] == '/') return b; return a + "/" + b; }

int main(int argc, char **argv) {
  if (argc < 2) { std::printf("❌ Invalid number of arguments. Please supply project-config.json as argument\n"); return 1; }
  if (!std::strcmp(argv[1], "create-template")) {
    const std::string t = config_template();
    if (!write_file("project-config-template.json", t.data(), t.size())) return 1;
    std::printf("✅ Written template object to project-config-template.json\n"); return 0;
  }
  int n_gpus = 1, device0 = 0, frames_per_batch = 32; bool force = false, encpy = false;
  for (int i = 2; i < argc; i++) {
    if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) n_gpus = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device0 = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--batch-frames") && i + 1 < argc) frames_per_batch = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--force")) force = true;
    else if (!std::strcmp(argv[i], "--encoder-py-manifest")) encpy = true;
  }
  std::vector<uint8_t> raw; if (!read_file(argv[1], raw)) { std::printf("❌ cannot read %s\n", argv[1]); return 1; }
  Config cfg; std::string err;
  if (!load_config(std::string(raw.begin(), raw.end()), cfg, err)) { std::printf("❌ %s\n", err.c_str()); return 1; }
  char cwd[4096]; if (!getcwd(cwd, sizeof cwd)) return 1;
  cfg.output_directory = join(cwd, cfg.output_directory);          // scripts/Encoder.py:201
  if (!make_dirs(cfg.output_directory)) { std::printf("❌ cannot create %s\n", cfg.output_directory.c_str()); return 1; }
  if (uvol_device_count() <= 0) { std::printf("❌ no HIP device: uvolenc has no CPU fallback\n"); return 1; }
  n_gpus = std::max(1, std::min(n_gpus, uvol_device_count() - device0));

  uvol_params prm; uvol_params_default(&prm);
  prm.q_position_attr = cfg.q_position; prm.q_texture_attr = cfg.q_texture; prm.q_normal_attr = cfg.q_normal; prm.q_generic_attr = cfg.q_generic;
  prm.draco_compression_level = cfg.compression_level; prm.ktx2_batch_size = cfg.ktx2_batch_size; prm.max_batch = frames_per_batch;
  std::vector<uvol_ctx *> ctxs((size_t)n_gpus, nullptr);
  for (int g = 0; g < n_gpus; g++) if (uvol_ctx_create(device0 + g, &prm, &ctxs[g]) != UVOL_OK) { std::printf("❌ cannot create codec context on GPU %d\n", device0 + g); return 1; }

  std::printf("🎯 Dealing with Geomety data\n");
  if (!cfg.abc_file_path.empty()) { std::printf("❌ ABCFilePath needs Blender (bpy); export OBJ files and use OBJFilesPath\n"); return 1; }
  const std::string geo_dir = join(cfg.output_directory, "geometry_draco");
  int pad = 5;
  if (!cfg.obj_files_path.empty()) {
    std::printf("🚧 Obtained OBJ files path\n");
    const std::string dir = dirname_of(cfg.obj_files_path), pat = basename_of(cfg.obj_files_path);
    { int h = (int)std::count(pat.begin(), pat.end(), '#'); if (h > 0) pad = h; }
    std::vector<std::string> files; for (auto &f : list_dir(dir)) if (match_pattern_lenient(pat, f)) files.push_back(f);
    if (!make_dirs(geo_dir)) return 1;
    std::atomic<int> failed{-1};
    // contiguous blocks of frames per GPU, each GPU encodes batches of frames_per_batch frames
    std::vector<std::thread> th;
    for (int g = 0; g < n_gpus; g++) th.emplace_back([&, g] {
      const size_t lo = files.size() * (size_t)g / n_gpus, hi = files.size() * (size_t)(g + 1) / n_gpus;
      for (size_t b0 = lo; b0 < hi && failed < 0; b0 += (size_t)frames_per_batch) {
        const size_t nb = std::min(hi - b0, (size_t)frames_per_batch);
        std::vector<ObjMesh> ms(nb); std::vector<uvol_mesh> um(nb); std::vector<std::vector<uint8_t>> outs(nb);
        std::vector<uint8_t *> op(nb); std::vector<size_t> caps(nb), lens(nb); std::vector<int> st(nb);
        for (size_t k = 0; k < nb; k++) {
          std::string e; if (!read_obj(join(dir, files[b0 + k]), ms[k], e)) { std::printf("Failed to compress %s\n%s\n", files[b0 + k].c_str(), e.c_str()); failed = (int)(b0 + k); return; }
          uvol_mesh &m = um[k]; std::memset(&m, 0, sizeof m);
          m.pos = ms[k].pos.data(); m.n_pos = (uint32_t)ms[k].pos.size() / 3; m.idx_pos = ms[k].idx_pos.data(); m.n_faces = (uint32_t)ms[k].idx_pos.size() / 3;
          if (!ms[k].uv.empty()) { m.uv = ms[k].uv.data(); m.n_uv = (uint32_t)ms[k].uv.size() / 2; m.idx_uv = ms[k].idx_uv.data(); }
          if (!ms[k].nrm.empty()) { m.nrm = ms[k].nrm.data(); m.n_nrm = (uint32_t)ms[k].nrm.size() / 3; m.idx_nrm = ms[k].idx_nrm.data(); }
          caps[k] = uvol_mesh_bound(&m); outs[k].resize(caps[k]); op[k] = outs[k].data();
        }
        if (uvol_encode_mesh_batch(ctxs[g], um.data(), (int)nb, op.data(), caps.data(), lens.data(), st.data()) != UVOL_OK) { std::printf("Failed to compress %s\n%s\n", files[b0].c_str(), uvol_last_error(ctxs[g])); failed = (int)b0; return; }
        for (size_t k = 0; k < nb; k++) {
          if (st[k] != UVOL_OK) { std::printf("Failed to compress %s\n", files[b0 + k].c_str()); failed = (int)(b0 + k); return; }   // scripts/Encoder.py:263-266
          char name[64]; std::snprintf(name, sizeof name, "%0*zu.drc", pad, b0 + k);
          if (!write_file(join(geo_dir, name), outs[k].data(), lens[k])) { failed = (int)(b0 + k); return; }
        }
      }
    });
    for (auto &t : th) t.join();
    if (failed >= 0) return 1;
    cfg.draco_files_path = join(geo_dir, std::string((size_t)pad, '#') + ".drc");
  }
  if (!cfg.draco_files_path.empty()) std::printf("✅ Obtained DRACO files\n");

  std::printf("🎯 Dealing with Texture data\n");
  const std::string tex_dir = join(cfg.output_directory, "texture_ktx2_baseColor_default");
  uint32_t tex_w = 0, tex_h = 0;
  if (!cfg.images_path.empty()) {
    std::printf("🚧 Obtained Images path.\n");
    const std::string cpat = convert_pounds_to_c_style(cfg.images_path);     // scripts/Encoder.py:274
    if (!make_dirs(tex_dir)) return 1;
    std::vector<int> starts; for (int i = cfg.ktx2_first_file; i < cfg.ktx2_file_count; i += cfg.ktx2_batch_size) starts.push_back(i);   // :282-287
    std::atomic<int> failed{-1};
    std::vector<std::thread> th;
    for (int g = 0; g < n_gpus; g++) th.emplace_back([&, g] {
      const size_t lo = starts.size() * (size_t)g / n_gpus, hi = starts.size() * (size_t)(g + 1) / n_gpus;
      for (size_t s = lo; s < hi && failed < 0; s++) {
        std::vector<Image> imgs; std::vector<const uint8_t *> ptrs;
        for (int k = 0; k < cfg.ktx2_batch_size; k++) {
          char path[4096]; std::snprintf(path, sizeof path, cpat.c_str(), (unsigned)(starts[s] + k));
          Image im; std::string e;
          if (!read_png(path, im, e)) { if (k == 0 || starts[s] + k < cfg.ktx2_file_count) { std::printf("Failed to compress images with indices: [%d, %d]\n%s\n", starts[s], starts[s] + cfg.ktx2_batch_size, e.c_str()); failed = starts[s]; return; } break; }
          if (!imgs.empty() && (im.w != imgs[0].w || im.h != imgs[0].h)) { std::printf("Failed to compress images with indices: [%d, %d]\nimage sizes differ\n", starts[s], starts[s] + cfg.ktx2_batch_size); failed = starts[s]; return; }
          imgs.push_back(std::move(im));
        }
        for (auto &im : imgs) ptrs.push_back(im.rgba.data());
        tex_w = imgs[0].w; tex_h = imgs[0].h;
        std::vector<uint8_t> out(uvol_texture_bound(tex_w, tex_h, (int)imgs.size())); size_t len = 0;
        if (uvol_encode_texture_segment(ctxs[g], ptrs.data(), (int)ptrs.size(), tex_w, tex_h, out.data(), out.size(), &len) != UVOL_OK) {
          std::printf("Failed to compress images with indices: [%d, %d]\n%s\n", starts[s], starts[s] + cfg.ktx2_batch_size, uvol_last_error(ctxs[g])); failed = starts[s]; return; }   // :293-298
        char name[64]; std::snprintf(name, sizeof name, "%0*d.ktx2", pad, (starts[s] - cfg.ktx2_first_file) / cfg.ktx2_batch_size);
        if (!write_file(join(tex_dir, name), out.data(), len)) { failed = starts[s]; return; }
      }
    });
    for (auto &t : th) t.join();
    if (failed >= 0) return 1;
    cfg.ktx2_files_path = join(tex_dir, std::string((size_t)pad, '#') + ".ktx2");
  }
  if (!cfg.ktx2_files_path.empty()) std::printf("✅ Obtained KTX2 files\n");
  for (auto *c : ctxs) uvol_ctx_destroy(c);

  FrameCounts fc;
  if (!check_total_frames(cfg.draco_files_path, cfg.ktx2_files_path, cfg.ktx2_batch_size, cfg.geometry_frame_rate, cfg.texture_frame_rate, fc, err)) { std::printf("❌ %s\n", err.c_str()); return 1; }
  std::printf("Geometry frame count: %ld\nTexture frame count (not segments): %ld\n", fc.geometry_frames, fc.texture_frames);
  if (!fc.compatible) {
    std::printf("❌ Number of Geometry frames and Texture frames are not compatible with the given frame rates\n");
    if (!force) { std::printf("(re-run with --force to proceed anyway)\n"); return 1; }        // the reference prompts y/n (:141-146)
  } else std::printf("✅ Frames and frame rates are compatible\n");
  if (!tex_w) { std::vector<uint8_t> k; std::string d = dirname_of(cfg.ktx2_files_path); for (auto &f : list_dir(d)) if (match_pattern_lenient(basename_of(cfg.ktx2_files_path), f)) { if (read_file(join(d, f), k) && k.size() >= 28) { std::memcpy(&tex_w, &k[20], 4); std::memcpy(&tex_h, &k[24], 4); } break; } }
  const std::string man = json_dump(manifest_player(cfg, fc.geometry_frames, fc.texture_segments, tex_w, tex_h, pad));
  const std::string mpath = join(cfg.output_directory, "uvol.json");
  if (!write_file(mpath, man.data(), man.size())) return 1;
  if (encpy) { const std::string m2 = json_dump(manifest_encoder_py(cfg, fc.geometry_frames, fc.texture_segments, "geometry_draco/" + std::string((size_t)pad, '#') + ".drc", "texture_ktx2_baseColor_default/" + std::string((size_t)pad, '#') + ".ktx2"));
    write_file(join(cfg.output_directory, "uvol.encoderpy.json"), m2.data(), m2.size()); }
  std::printf("✅ Written Manifest file: %s.\n", mpath.c_str());
  return 0;
}
