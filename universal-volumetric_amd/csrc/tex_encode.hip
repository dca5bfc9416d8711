// tex_encode.hip — placeholder until the ETC1S/BasisLZ pipeline lands (same round).
#include "uvol_common.hpp"
struct TexState { int dummy; };
int tex_create(uvol_ctx *ctx) { ctx->tex = new TexState(); return UVOL_OK; }
void tex_destroy(uvol_ctx *ctx) { delete ctx->tex; ctx->tex = nullptr; }
size_t uvol_texture_bound(uint32_t w, uint32_t h, int n) { return 65536 + (size_t)((w + 3) / 4) * ((h + 3) / 4) * 8 * (size_t)(n > 0 ? n : 1); }
int tex_encode_segment(uvol_ctx *ctx, const uint8_t *const *, int, uint32_t, uint32_t, bool, uint8_t *, size_t, size_t *) {
  ctx->set_error("texture path not built yet"); return UVOL_E_UNSUPPORTED;
}
