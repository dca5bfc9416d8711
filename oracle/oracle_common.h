/* oracle/oracle_common.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Shared primitives of the CPU oracle (plain C restatement of the UVOL hot path:
 * Draco 2.2 mesh bitstream + KTX2/BasisLZ ETC1S).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link this.  The product library under
 * universal-volumetric_amd/ never includes or links anything in oracle/.
 *
 * The algorithm lives in third-party code that is NOT under /root/reference
 * (google/draco — decoder pinned 1.4.3 by src/V2/player.ts:101; BinomialLLC/basis_universal
 * "Basis Universal 1.16" per the fixtures' KTXwriter key).  It is restated from the published
 * bitstream behaviour recorded in SURVEY.md Appendix A/B/D and pinned against the reference's own
 * fixtures example/public/liam/output/{geometry_draco/NNNNN.drc, texture_ktx2-.../NNNNN.ktx2}.
 */
#ifndef UVOL_ORACLE_COMMON_H
#define UVOL_ORACLE_COMMON_H
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ORC_INV (-1)

/* ---- growable byte buffer ---- */
typedef struct { uint8_t *p; size_t n, cap; } orc_buf;
static inline void ob_reserve(orc_buf *b, size_t extra) {
  if (b->n + extra > b->cap) {
    size_t nc = b->cap ? b->cap * 2 : 4096;
    while (nc < b->n + extra) nc *= 2;
    b->p = (uint8_t *)realloc(b->p, nc); b->cap = nc;
  }
}
static inline void ob_u8(orc_buf *b, uint8_t v) { ob_reserve(b, 1); b->p[b->n++] = v; }
static inline void ob_bytes(orc_buf *b, const void *s, size_t n) { ob_reserve(b, n); memcpy(b->p + b->n, s, n); b->n += n; }
static inline void ob_u16(orc_buf *b, uint16_t v) { ob_bytes(b, &v, 2); }
static inline void ob_u32(orc_buf *b, uint32_t v) { ob_bytes(b, &v, 4); }
static inline void ob_i32(orc_buf *b, int32_t v) { ob_bytes(b, &v, 4); }
static inline void ob_u64(orc_buf *b, uint64_t v) { ob_bytes(b, &v, 8); }
static inline void ob_f32(orc_buf *b, float v) { ob_bytes(b, &v, 4); }
static inline void ob_varint(orc_buf *b, uint64_t v) {
  while (v >= 0x80) { ob_u8(b, (uint8_t)(v | 0x80)); v >>= 7; }
  ob_u8(b, (uint8_t)v);
}
static inline void ob_free(orc_buf *b) { free(b->p); b->p = NULL; b->n = b->cap = 0; }

/* ---- corner helpers ---- */
static inline int c_nxt(int c) { return (c % 3 == 2) ? c - 2 : c + 1; }
static inline int c_prv(int c) { return (c % 3 == 0) ? c + 2 : c - 1; }

/* C truncating division is the C `/` operator on int64. */
static inline uint64_t orc_isqrt(uint64_t n) {
  if (n == 0) return 0;
  uint64_t a = n, r = 1;
  while (a >= 2) { r *= 2; a /= 4; }
  do { r = (r + n / r) / 2; } while (r * r > n);
  return r;
}

uint32_t orc_crc32(const void *data, size_t n);

#endif
