// basisu_shim.cpp — argv-compatible stand-in for the `basisu` process the stock driver spawns:
//   basisu -ktx2 -tex_type video -multifile_printf P -multifile_num B -multifile_first i -y_flip -output_file out.ktx2   (scripts/Encoder.py:290)
// The ETC1S / KTX2 / video path the reference uses, and -uastc (UASTC LDR 4x4 blocks in the same container); unknown flags are ignored.
#include "uvol_host.hpp"
#include "../../include/uvol_codec.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, char **argv) {
  std::string pat, out; int num = 1, first = 0; bool ktx2 = false; std::vector<std::string> files;
  uvol_params prm; uvol_params_default(&prm); prm.y_flip = 0;
  for (int i = 1; i < argc; i++) {
    auto val = [&]() { return i + 1 < argc ? argv[++i] : ""; };
    if (!std::strcmp(argv[i], "-ktx2")) ktx2 = true; else if (!std::strcmp(argv[i], "-y_flip")) prm.y_flip = 1;
    else if (!std::strcmp(argv[i], "-tex_type")) (void)val(); else if (!std::strcmp(argv[i], "-multifile_printf")) pat = val();
    else if (!std::strcmp(argv[i], "-multifile_num")) num = std::atoi(val()); else if (!std::strcmp(argv[i], "-multifile_first")) first = std::atoi(val());
    else if (!std::strcmp(argv[i], "-output_file")) out = val(); else if (!std::strcmp(argv[i], "-q")) prm.etc1s_quality = std::atoi(val());
    else if (!std::strcmp(argv[i], "-file")) files.push_back(val());
    else if (!std::strcmp(argv[i], "-uastc")) prm.uastc = 1;                      // UASTC LDR 4x4 mode (KTX2 without Zstandard supercompression)
    else if (argv[i][0] != '-') files.push_back(argv[i]);
  }
  if (!ktx2 || out.empty()) { std::fprintf(stderr, "basisu (uvol shim): expected -ktx2 ... -output_file <path>\n"); return 1; }
  if (!pat.empty()) for (int k = 0; k < num; k++) { char p[4096]; std::snprintf(p, sizeof p, pat.c_str(), (unsigned)(first + k)); files.push_back(p); }
  if (files.empty()) { std::fprintf(stderr, "No input files\n"); return 1; }
  std::vector<uvolh::Image> imgs(files.size()); std::vector<const uint8_t *> ptrs; std::string err;
  for (size_t k = 0; k < files.size(); k++) {
    if (!uvolh::read_png(files[k], imgs[k], err)) { std::fprintf(stderr, "Failed reading source image: %s\n", err.c_str()); return 1; }
    if (imgs[k].w != imgs[0].w || imgs[k].h != imgs[0].h) { std::fprintf(stderr, "All source images must have the same dimensions\n"); return 1; }
    ptrs.push_back(imgs[k].rgba.data());
  }
  prm.ktx2_batch_size = (int)files.size();
  uvol_ctx *ctx = nullptr;
  if (uvol_ctx_create(0, &prm, &ctx) != UVOL_OK) { std::fprintf(stderr, "basisu (uvol shim): no HIP device, no CPU fallback\n"); return 2; }
  std::vector<uint8_t> buf(uvol_texture_bound(imgs[0].w, imgs[0].h, (int)files.size())); size_t len = 0;
  const int rc = uvol_encode_texture_segment(ctx, ptrs.data(), (int)ptrs.size(), imgs[0].w, imgs[0].h, buf.data(), buf.size(), &len);
  if (rc != UVOL_OK) { std::fprintf(stderr, "Compression failed: %s\n", uvol_last_error(ctx)); uvol_ctx_destroy(ctx); return 3; }
  uvol_ctx_destroy(ctx);
  if (!uvolh::write_file(out, buf.data(), len)) { std::fprintf(stderr, "Failed writing output file\n"); return 4; }
  std::printf("Wrote %s (%zu bytes, %zu layers)\n", out.c_str(), len, files.size());
  return 0;
}
