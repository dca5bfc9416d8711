/* oracle/ktx2_oracle.h — TEST INFRASTRUCTURE (see oracle_common.h).
 * CPU oracle for KTX2 + BasisLZ/ETC1S as produced by
 * `basisu -ktx2 -tex_type video -multifile_* -y_flip` (scripts/Encoder.py:290) and pinned by the
 * reference fixtures example/public/liam/output/texture_ktx2-fps30-1k_baseColor_default/NNNNN.ktx2
 * (SURVEY.md Appendix B / C).  Third-party origin: BinomialLLC/basis_universal 1.16 (not vendored).
 */
#ifndef UVOL_KTX2_ORACLE_H
#define UVOL_KTX2_ORACLE_H
#include "oracle_common.h"
#ifdef __cplusplus
extern "C" {
#endif

#define KTX2_MAX_LAYERS 64

typedef struct {
  uint32_t vk_format, type_size, width, height, depth, layers, faces, levels, supercomp;
  uint32_t dfd_off, dfd_len, kvd_off, kvd_len; uint64_t sgd_off, sgd_len;
  uint64_t level_off, level_len, level_ulen;
  uint32_t dfd_model, dfd_transfer, dfd_primaries;
  uint32_t n_endpoints, n_selectors, endpoints_len, selectors_len, tables_len, extended_len;
  uint32_t bx, by;
  uint8_t *endpoints;            /* n_endpoints * 4: r5 g5 b5 inten */
  uint32_t *selectors;           /* n_selectors: byte j = row j, texel x at bits 2x..2x+1 (0..3 low->high) */
  uint32_t hist_size;
  uint32_t ep_bits_used, sel_bits_used, tab_bits_used;
  int n_slices;
  uint32_t slice_flags[KTX2_MAX_LAYERS], slice_off[KTX2_MAX_LAYERS], slice_len[KTX2_MAX_LAYERS];
  uint64_t slice_bits_used[KTX2_MAX_LAYERS];
  uint32_t slice_skip[KTX2_MAX_LAYERS];
  uint16_t *block_ei, *block_si; /* n_slices * bx*by */
  char writer[64];
  uint32_t anim_duration, anim_timescale, anim_loops; int has_anim;
  int has_alpha;                 /* every image has a second (alpha) slice: slice 2 * layer + 1; n_slices = 2 * layers */
} ktx2_file;

int ktx2_decode(const uint8_t *b, size_t n, ktx2_file *f);
void ktx2_free(ktx2_file *f);
/* decode one layer (image) to RGBA8, rows in stored order (top of stored image first); alpha from the image's alpha slice (its green
 * channel, as the basis transcoder does), 255 without one */
void ktx2_layer_rgba(const ktx2_file *f, int layer, uint8_t *out);

/* ---- encoder restatement ---- */
typedef struct {
  int quality;      /* 1..255, default 128 */
  int y_flip;       /* default 1 (scripts/Encoder.py:290 passes -y_flip) */
} ktx2_enc_params;
/* layers: n_layers pointers to width*height*4 RGBA8 (top row first). Output: .ktx2 bytes. */
int ktx2_encode(const uint8_t *const *layers, int n_layers, uint32_t width, uint32_t height,
                const ktx2_enc_params *p, orc_buf *out);

#ifdef __cplusplus
}
#endif
#endif
