/* oracle/uastc_oracle.h — TEST INFRASTRUCTURE (see oracle_common.h, uastc.c).
 * CPU oracle of the UASTC LDR 4x4 texture mode and of its ASTC 4x4 transcode (BASELINE configs[4]).
 * PARITY UNPINNED against basisu / the basis transcoder: restated from the published block formats, see uastc.c. */
#ifndef UVOL_UASTC_ORACLE_H
#define UVOL_UASTC_ORACLE_H
#include "oracle_common.h"
#ifdef __cplusplus
extern "C" {
#endif

/* logical content of one UASTC block (single-subset modes) */
typedef struct {
  int mode;                 /* 0..18 */
  int ccs;                  /* component of the second weight plane (dual-plane modes) */
  uint8_t ep[8];            /* endpoint codes in ASTC order: c0.lo c0.hi c1.lo c1.hi ...; code = (trit|quint) << bits | bits */
  uint8_t w[32];            /* weight indices, raster order, planes interleaved */
  uint8_t solid[4];         /* mode 8 */
  int bc1_hint0, bc1_hint1, etc1_flip, etc1_diff, etc1_inten0, etc1_inten1, etc1_bias, etc2_hints, etc1_sel;
  uint8_t etc1_base[3];     /* mode 8: 5-bit ETC1 base colour */
} uastc_lblock;

int uastc_mode_emitted(int mode);
int astc_unquant_endpoint(int range, int code);
int astc_unquant_weight_bits(int bits, int index);
void uastc_pack(const uastc_lblock *L, uint8_t out[16]);
int uastc_unpack(const uint8_t in[16], uastc_lblock *L);
void uastc_lblock_rgba(const uastc_lblock *L, uint8_t rgba[64]);
int uastc_decode_block(const uint8_t in[16], uint8_t rgba[64]);          /* texel i = 4*y + x */
int uastc_to_astc(const uint8_t in[16], uint8_t out[16]);
int astc_decode_block(const uint8_t in[16], uint8_t rgba[64]);           /* independent decoder: 4x4 grid, 1 partition, CEM 8 / 12, void extent */
void uastc_encode_lblock(const uint8_t px[64], uastc_lblock *out);
void uastc_encode_block(const uint8_t px[64], uint8_t out[16]);

/* n_layers RGBA8 images (top row first) -> KTX2 (vkFormat 0, DFD model 166, no supercompression, layers as array layers) */
int uastc_ktx2_encode(const uint8_t *const *layers, int n_layers, uint32_t width, uint32_t height, int y_flip, orc_buf *out);
int uastc_ktx2_info(const uint8_t *b, size_t n, uint32_t *width, uint32_t *height, uint32_t *layers, uint64_t *level_off, int *has_alpha);
/* target 0: RGBA8 (layers * width * height * 4, stored row order), target 1: ASTC 4x4 blocks (layers * bx * by * 16) */
int uastc_ktx2_decode(const uint8_t *b, size_t n, int target, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
