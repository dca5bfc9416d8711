// draco_encoder_shim.cpp — argv-compatible stand-in for the `draco_encoder` process the stock driver spawns:
//   draco_encoder -i f.obj -o f.drc -qp 11 -qt 10 -qn 8 -qg 8 -cl 7        (scripts/Encoder.py:260)
// Exit code 0 on success, non-zero otherwise (the caller only checks rc, :262-266); stdout is ignored by the caller.
#include "uvol_host.hpp"
#include "../../include/uvol_codec.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, char **argv) {
  std::string in, out; uvol_params prm; uvol_params_default(&prm);
  for (int i = 1; i < argc; i++) {
    auto val = [&]() { return i + 1 < argc ? argv[++i] : ""; };
    if (!std::strcmp(argv[i], "-i")) in = val(); else if (!std::strcmp(argv[i], "-o")) out = val();
    else if (!std::strcmp(argv[i], "-qp")) prm.q_position_attr = std::atoi(val()); else if (!std::strcmp(argv[i], "-qt")) prm.q_texture_attr = std::atoi(val());
    else if (!std::strcmp(argv[i], "-qn")) prm.q_normal_attr = std::atoi(val()); else if (!std::strcmp(argv[i], "-qg")) prm.q_generic_attr = std::atoi(val());
    else if (!std::strcmp(argv[i], "-cl")) prm.draco_compression_level = std::atoi(val());
    else if (argv[i][0] == '-' && i + 1 < argc && argv[i + 1][0] != '-') ++i;          // unknown option with a value: ignored
  }
  if (in.empty()) { std::fprintf(stderr, "Usage: draco_encoder -i <input.obj> -o <output.drc> [-qp -qt -qn -qg -cl]\n"); return 1; }
  if (out.empty()) out = in + ".drc";
  if (prm.draco_compression_level < 0 || prm.draco_compression_level > 10) { std::fprintf(stderr, "Error: The compression level must be in [0, 10].\n"); return 1; }
  if (prm.draco_compression_level == 0) std::fprintf(stderr, "draco_encoder (uvol shim): -cl 0 = sequential connectivity with the difference predictor (what stock draco_encoder selects at this level)\n");
  else if (prm.draco_compression_level != 7) std::fprintf(stderr, "draco_encoder (uvol shim): -cl %d is encoded with the cl 7 tool set (same bitstream syntax, any Draco decoder reads it)\n", prm.draco_compression_level);
  uvolh::ObjMesh m; std::string err;
  if (!uvolh::read_obj(in, m, err)) { std::fprintf(stderr, "Failed loading the input mesh: %s\n", err.c_str()); return 1; }
  uvol_ctx *ctx = nullptr;
  if (uvol_ctx_create(0, &prm, &ctx) != UVOL_OK) { std::fprintf(stderr, "draco_encoder (uvol shim): no HIP device, no CPU fallback\n"); return 2; }
  uvol_mesh um; std::memset(&um, 0, sizeof um);
  um.pos = m.pos.data(); um.n_pos = (uint32_t)m.pos.size() / 3; um.idx_pos = m.idx_pos.data(); um.n_faces = (uint32_t)m.idx_pos.size() / 3;
  if (!m.uv.empty()) { um.uv = m.uv.data(); um.n_uv = (uint32_t)m.uv.size() / 2; um.idx_uv = m.idx_uv.data(); }
  if (!m.nrm.empty()) { um.nrm = m.nrm.data(); um.n_nrm = (uint32_t)m.nrm.size() / 3; um.idx_nrm = m.idx_nrm.data(); }
  std::vector<uint8_t> buf(uvol_mesh_bound(&um)); size_t len = 0;
  const int rc = uvol_encode_mesh(ctx, &um, buf.data(), buf.size(), &len);
  if (rc != UVOL_OK) { std::fprintf(stderr, "Failed to encode the mesh: %s\n", uvol_last_error(ctx)); uvol_ctx_destroy(ctx); return 3; }
  uvol_ctx_destroy(ctx);
  if (!uvolh::write_file(out, buf.data(), len)) { std::fprintf(stderr, "Failed to create the output file.\n"); return 4; }
  std::printf("Encoded mesh saved to %s (%zu bytes)\n", out.c_str(), len);
  return 0;
}
