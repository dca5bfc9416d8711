/* oracle/uastc.c — TEST INFRASTRUCTURE (see oracle_common.h).
 *
 * CPU oracle of the UASTC LDR 4x4 texture mode (BASELINE north_star "BasisU ETC1S/UASTC 4x4-block endpoint/selector
 * search"; BASELINE configs[4] "KTX2 -> ASTC transcode"): the block format `basisu -uastc -ktx2` writes, its decode to RGBA8, and
 * the UASTC -> ASTC 4x4 transcode the stock player asks its transcoder for first (reference src/lib/KTX2Loader.js:591-600: for
 * UASTC sources ASTC_4x4 has priority 1; :648-689 picks the first supported target).
 *
 * Where the arithmetic lives: BinomialLLC/basis_universal (encoder unpinned, scripts/Encoder.py:37; transcoder pinned to
 * three@0.153.0 by src/V2/player.ts:97) — NOT under /root/reference, not installed, no network.  The reference driver never
 * passes -uastc (scripts/Encoder.py:290) and the reference holds NO UASTC fixture.  This file therefore restates the PUBLISHED
 * formats from the specifications:
 *   - "UASTC LDR 4x4 Texture Specification" (basis_universal wiki): 19 modes, mode prefix codes, hint fields, endpoint packing
 *     (base-3 / base-5 bundles first, then the plain bits), weights with the anchor's top bit dropped;
 *   - ASTC LDR profile (Khronos Data Format Specification, ASTC chapter): block mode, CEM 8 / 12, integer sequence encoding,
 *     endpoint / weight unquantisation, weight bit reversal, void-extent blocks.
 * PARITY UNPINNED: nothing here has been compared with basisu or with the basis transcoder.  What pins it is internal: the
 * per-mode field widths restated here add up to exactly 128 bits for modes 0, 6, 10, 11, 12, 16, 18, the 19 + 1 mode prefix
 * codes form a complete prefix code (Kraft sum 1), the ASTC block-mode words decode to the weight grids / ranges the modes
 * state, the endpoint range of every mode is the range an ASTC decoder infers from the bits left over — and the two
 * independently written decoders below (UASTC -> RGBA, ASTC -> RGBA through a generic integer-sequence decoder) agree on every
 * block the encoder emits (tests/test_oracle_uastc.py).
 *
 * Only single-subset modes are EMITTED (0, 6, 18 for opaque blocks, 10, 11, 12 for blocks with alpha, 8 for solid blocks):
 * those are the modes whose bit layout is fully determined by the field widths (they fill 128 bits exactly), and the 2- / 3-
 * subset modes need basisu's tables of common ASTC/BC7 partition patterns, which are not restated.  Any UASTC decoder accepts a
 * stream that uses a subset of the modes.
 */
#include "uastc_oracle.h"
#include <stdio.h>

/* ---------------------------------------------------------------- tables */
/* ASTC quantisation ranges 0..20: levels, and bits / trits / quints of the integer sequence encoding */
static const uint16_t R_LEVELS[21] = { 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192, 256 };
static const uint8_t R_BTQ[21][3] = { {1,0,0}, {0,1,0}, {2,0,0}, {0,0,1}, {1,1,0}, {3,0,0}, {1,0,1}, {2,1,0}, {4,0,0}, {2,0,1}, {3,1,0},
                                      {5,0,0}, {3,0,1}, {4,1,0}, {6,0,0}, {4,0,1}, {5,1,0}, {7,0,0}, {5,0,1}, {6,1,0}, {8,0,0} };
/* UASTC modes (spec tables): prefix code {value, length} read LSB first, weight bits, endpoint range, components, planes, CEM */
static const uint8_t UM_HUFF[20][2] = { {0x1,4}, {0x35,6}, {0x1D,5}, {0x3,5}, {0x13,5}, {0xB,5}, {0x1B,5}, {0x7,5}, {0x17,5}, {0xF,5},
                                        {0x2,3}, {0x0,2}, {0x6,3}, {0x1F,5}, {0xD,5}, {0x5,7}, {0x15,6}, {0x25,6}, {0x9,4}, {0x45,7} };
static const uint8_t UM_WBITS[19] = { 4, 2, 3, 2, 2, 3, 2, 2, 0, 2, 4, 2, 3, 1, 2, 4, 2, 2, 5 };
static const uint8_t UM_RANGE[19] = { 19, 20, 8, 7, 12, 20, 18, 12, 0, 8, 13, 13, 19, 20, 20, 20, 20, 20, 11 };
static const uint8_t UM_COMPS[19] = { 3, 3, 3, 3, 3, 3, 3, 3, 0, 4, 4, 4, 4, 4, 4, 2, 2, 2, 3 };
static const uint8_t UM_PLANES[19] = { 1, 1, 1, 1, 1, 1, 2, 1, 0, 1, 1, 2, 1, 2, 1, 1, 1, 2, 1 };
static const uint8_t UM_SUBSETS[19] = { 1, 1, 2, 3, 2, 1, 1, 2, 0, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1 };
static const uint8_t UM_BIAS[19] = { 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1 };
static const uint8_t UM_BC1H0[19] = { 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1 };
static const uint8_t UM_BC1H1[19] = { 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1 };
static const uint8_t UM_ALPHA[19] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0 };
/* the ASTC block-mode word of every mode (4x4 weight grid, weight range, dual-plane bit) */
static const uint16_t UM_ASTC_BM[19] = { 0x242, 0x42, 0x53, 0x42, 0x42, 0x53, 0x442, 0x42, 0, 0x42, 0x242, 0x442, 0x53, 0x441, 0x42, 0x242, 0x42, 0x442, 0x253 };

int uastc_mode_emitted(int m) { return m == 0 || m == 6 || m == 8 || m == 10 || m == 11 || m == 12 || m == 18; }

/* ---------------------------------------------------------------- unquantisation (ASTC spec, "Endpoint Unquantization") */
int astc_unquant_endpoint(int range, int v) {
  const int bits = R_BTQ[range][0], tr = R_BTQ[range][1], qu = R_BTQ[range][2];
  if (!tr && !qu) {                                   /* bit replication to 8 bits */
    int r = 0, have = 0;
    while (have < 8) { r = (r << bits) | v; have += bits; }
    return (r >> (have - 8)) & 255;
  }
  const int D = v >> bits, m = v & ((1 << bits) - 1);
  const int a = m & 1, b = (m >> 1) & 1, c = (m >> 2) & 1, d = (m >> 3) & 1, e = (m >> 4) & 1, f = (m >> 5) & 1;
  const int A = a ? 0x1FF : 0;
  int B = 0, C = 0;
  if (tr) {
    switch (bits) {
      case 1: C = 204; B = 0; break;
      case 2: C = 93; B = (b << 8) | (b << 4) | (b << 2) | (b << 1); break;                                   /* b000b0bb0 */
      case 3: C = 44; B = (c << 8) | (b << 7) | (c << 3) | (b << 2) | (c << 1) | b; break;                    /* cb000cbcb */
      case 4: C = 22; B = (d << 8) | (c << 7) | (b << 6) | (d << 2) | (c << 1) | b; break;                    /* dcb000dcb */
      case 5: C = 11; B = (e << 8) | (d << 7) | (c << 6) | (b << 5) | (e << 1) | d; break;                    /* edcb000ed */
      default: C = 5; B = (f << 8) | (e << 7) | (d << 6) | (c << 5) | (b << 4) | f; break;                    /* fedcb000f */
    }
  } else {
    switch (bits) {
      case 1: C = 113; B = 0; break;
      case 2: C = 54; B = (b << 8) | (b << 3) | (b << 2); break;                                              /* b0000bb00 */
      case 3: C = 26; B = (c << 8) | (b << 7) | (c << 2) | (b << 1) | c; break;                               /* cb0000cbc */
      case 4: C = 13; B = (d << 8) | (c << 7) | (b << 6) | (d << 1) | c; break;                               /* dcb0000dc */
      default: C = 6; B = (e << 8) | (d << 7) | (c << 6) | (b << 5) | e; break;                               /* edcb0000e */
    }
  }
  int T = D * C + B;
  T ^= A;
  return (A & 0x80) | (T >> 2);
}
/* weights: only the bit-only ranges occur here (1..5 bits): replicate to 6 bits, values above 32 move up by one (0..64) */
int astc_unquant_weight_bits(int bits, int v) {
  int r = 0, have = 0;
  while (have < 6) { r = (r << bits) | v; have += bits; }
  r = (r >> (have - 6)) & 63;
  return r > 32 ? r + 1 : r;
}
static inline int astc_interp(int l, int h, int w) {          /* LDR, non-sRGB decode mode: 8 -> 16 bits by replication */
  l = (l << 8) | l; h = (h << 8) | h;
  return ((l * (64 - w) + h * w + 32) >> 6) >> 8;
}

/* ---------------------------------------------------------------- LSB-first bit I/O on a 16-byte block */
static void put_bits(uint8_t *b, int *o, uint32_t v, int n) { for (int i = 0; i < n; i++, (*o)++) if ((v >> i) & 1) b[*o >> 3] |= (uint8_t)(1u << (*o & 7)); }
static uint32_t get_bits(const uint8_t *b, int *o, int n) { uint32_t v = 0; for (int i = 0; i < n; i++, (*o)++) v |= (uint32_t)((b[*o >> 3] >> (*o & 7)) & 1) << i; return v; }

/* ---------------------------------------------------------------- UASTC block pack / unpack */
static int mode_of_prefix(const uint8_t *b, int *len) {
  for (int m = 0; m < 19; m++) { const int l = UM_HUFF[m][1]; int o = 0; if (get_bits(b, &o, l) == UM_HUFF[m][0]) { *len = l; return m; } }
  return -1;
}

void uastc_pack(const uastc_lblock *L, uint8_t out[16]) {
  memset(out, 0, 16);
  int o = 0; const int m = L->mode;
  put_bits(out, &o, UM_HUFF[m][0], UM_HUFF[m][1]);
  if (m == 8) {
    for (int c = 0; c < 4; c++) put_bits(out, &o, L->solid[c], 8);
    put_bits(out, &o, (uint32_t)L->etc1_diff, 1); put_bits(out, &o, (uint32_t)L->etc1_inten0, 3); put_bits(out, &o, (uint32_t)L->etc1_sel, 2);
    for (int c = 0; c < 3; c++) put_bits(out, &o, L->etc1_base[c], 5);
    return;
  }
  if (UM_BC1H0[m]) put_bits(out, &o, (uint32_t)L->bc1_hint0, 1);
  if (UM_BC1H1[m]) put_bits(out, &o, (uint32_t)L->bc1_hint1, 1);
  put_bits(out, &o, (uint32_t)L->etc1_flip, 1); put_bits(out, &o, (uint32_t)L->etc1_diff, 1);
  put_bits(out, &o, (uint32_t)L->etc1_inten0, 3); put_bits(out, &o, (uint32_t)L->etc1_inten1, 3);
  if (UM_BIAS[m]) put_bits(out, &o, (uint32_t)L->etc1_bias, 5);
  if (UM_ALPHA[m]) put_bits(out, &o, (uint32_t)L->etc2_hints, 8);
  if (m == 6 || m == 11 || m == 13) put_bits(out, &o, (uint32_t)L->ccs, 2);
  /* endpoints: the base-3 / base-5 bundles first, then the plain low bits of every value */
  const int nv = 2 * UM_COMPS[m], range = UM_RANGE[m], bits = R_BTQ[range][0], tr = R_BTQ[range][1], qu = R_BTQ[range][2];
  if (tr || qu) {
    const int per = tr ? 5 : 3, mul = tr ? 3 : 5, groups = (nv + per - 1) / per;
    for (int g = 0; g < groups; g++) {
      uint32_t acc = 0, scale = 1; int cnt = 0;
      for (int k = 0; k < per && g * per + k < nv; k++, cnt++) { acc += (uint32_t)(L->ep[g * per + k] >> bits) * scale; scale *= (uint32_t)mul; }
      int nb = tr ? 8 : 7;
      if (cnt < per) { static const uint8_t tb[5] = { 0, 2, 4, 5, 7 }, qb[3] = { 0, 3, 5 }; nb = tr ? tb[cnt] : qb[cnt]; }
      put_bits(out, &o, acc, nb);
    }
  }
  for (int i = 0; i < nv; i++) put_bits(out, &o, (uint32_t)L->ep[i] & ((1u << bits) - 1u), bits);
  /* weights: plane-interleaved raster order; the first weight of every plane is stored without its (zero) top bit */
  const int planes = UM_PLANES[m], wb = UM_WBITS[m];
  for (int i = 0; i < 16 * planes; i++) put_bits(out, &o, L->w[i], i < planes ? wb - 1 : wb);
}

int uastc_unpack(const uint8_t in[16], uastc_lblock *L) {
  memset(L, 0, sizeof(*L));
  int o = 0, hl = 0; const int m = mode_of_prefix(in, &hl);
  if (m < 0) return -1;
  L->mode = m; o = hl;
  if (m == 8) {
    for (int c = 0; c < 4; c++) L->solid[c] = (uint8_t)get_bits(in, &o, 8);
    L->etc1_diff = (int)get_bits(in, &o, 1); L->etc1_inten0 = (int)get_bits(in, &o, 3); L->etc1_sel = (int)get_bits(in, &o, 2);
    for (int c = 0; c < 3; c++) L->etc1_base[c] = (uint8_t)get_bits(in, &o, 5);
    return 0;
  }
  if (UM_SUBSETS[m] != 1 || !uastc_mode_emitted(m)) return -2;          /* multi-subset modes / modes with slack bits: not restated */
  if (UM_BC1H0[m]) L->bc1_hint0 = (int)get_bits(in, &o, 1);
  if (UM_BC1H1[m]) L->bc1_hint1 = (int)get_bits(in, &o, 1);
  L->etc1_flip = (int)get_bits(in, &o, 1); L->etc1_diff = (int)get_bits(in, &o, 1);
  L->etc1_inten0 = (int)get_bits(in, &o, 3); L->etc1_inten1 = (int)get_bits(in, &o, 3);
  if (UM_BIAS[m]) L->etc1_bias = (int)get_bits(in, &o, 5);
  if (UM_ALPHA[m]) L->etc2_hints = (int)get_bits(in, &o, 8);
  if (m == 6 || m == 11 || m == 13) L->ccs = (int)get_bits(in, &o, 2);
  const int nv = 2 * UM_COMPS[m], range = UM_RANGE[m], bits = R_BTQ[range][0], tr = R_BTQ[range][1], qu = R_BTQ[range][2];
  uint32_t tq[8]; int groups = 0; const int per = tr ? 5 : 3, mul = tr ? 3 : 5;
  if (tr || qu) {
    groups = (nv + per - 1) / per;
    for (int g = 0; g < groups; g++) {
      const int cnt = nv - g * per < per ? nv - g * per : per; int nb = tr ? 8 : 7;
      if (cnt < per) { static const uint8_t tb[5] = { 0, 2, 4, 5, 7 }, qb[3] = { 0, 3, 5 }; nb = tr ? tb[cnt] : qb[cnt]; }
      tq[g] = get_bits(in, &o, nb);
    }
  }
  for (int i = 0; i < nv; i++) {
    uint32_t v = get_bits(in, &o, bits);
    if (groups) { uint32_t a = tq[i / per]; for (int k = 0; k < i % per; k++) a /= (uint32_t)mul; const uint32_t d = a % (uint32_t)mul; v |= d << bits; }
    if (v >= R_LEVELS[range]) return -3;
    L->ep[i] = (uint8_t)v;
  }
  const int planes = UM_PLANES[m], wb = UM_WBITS[m];
  for (int i = 0; i < 16 * planes; i++) L->w[i] = (uint8_t)get_bits(in, &o, i < planes ? wb - 1 : wb);
  return o == 128 ? 0 : -4;
}

/* logical block -> 16 RGBA texels.  UASTC endpoints are read in the stored order (no blue contraction: the ASTC transcode
 * below re-orders them where ASTC would otherwise apply it). */
void uastc_lblock_rgba(const uastc_lblock *L, uint8_t rgba[64]) {
  const int m = L->mode;
  if (m == 8) { for (int i = 0; i < 16; i++) memcpy(rgba + 4 * i, L->solid, 4); return; }
  const int nc = UM_COMPS[m], range = UM_RANGE[m], planes = UM_PLANES[m], wb = UM_WBITS[m];
  int lo[4] = { 0, 0, 0, 255 }, hi[4] = { 0, 0, 0, 255 };
  for (int c = 0; c < nc; c++) { lo[c] = astc_unquant_endpoint(range, L->ep[2 * c]); hi[c] = astc_unquant_endpoint(range, L->ep[2 * c + 1]); }
  for (int i = 0; i < 16; i++) {
    const int w0 = astc_unquant_weight_bits(wb, L->w[planes * i]), w1 = planes == 2 ? astc_unquant_weight_bits(wb, L->w[2 * i + 1]) : w0;
    for (int c = 0; c < 4; c++) rgba[4 * i + c] = (uint8_t)(c < nc ? astc_interp(lo[c], hi[c], (planes == 2 && c == L->ccs) ? w1 : w0) : 255);
  }
}
int uastc_decode_block(const uint8_t in[16], uint8_t rgba[64]) { uastc_lblock L; const int rc = uastc_unpack(in, &L); if (rc) return rc; uastc_lblock_rgba(&L, rgba); return 0; }

/* ---------------------------------------------------------------- ASTC integer sequence encoding (encoder side) */
/* T / Q words found by running the specification's DECODE equations over every word (lowest word per tuple wins, so the bits a
 * truncated last group drops are zero). */
static void trits_of(int T, int t[5]) {
  int C;
  if (((T >> 2) & 7) == 7) { C = ((T >> 5) << 2) | (T & 3); t[4] = t[3] = 2; }
  else { C = T & 31; if (((T >> 5) & 3) == 3) { t[4] = 2; t[3] = (T >> 7) & 1; } else { t[4] = (T >> 7) & 1; t[3] = (T >> 5) & 3; } }
  if ((C & 3) == 3) { t[2] = 2; t[1] = (C >> 4) & 1; t[0] = (((C >> 3) & 1) << 1) | (((C >> 2) & 1) & ~((C >> 3) & 1)); }
  else if (((C >> 2) & 3) == 3) { t[2] = 2; t[1] = 2; t[0] = C & 3; }
  else { t[2] = (C >> 4) & 1; t[1] = (C >> 2) & 3; t[0] = (((C >> 1) & 1) << 1) | ((C & 1) & ~((C >> 1) & 1)); }
}
static void quints_of(int Q, int q[3]) {
  if (((Q >> 1) & 3) == 3 && ((Q >> 5) & 3) == 0) { q[2] = ((Q & 1) << 2) | ((((Q >> 4) & 1) & ~(Q & 1)) << 1) | (((Q >> 3) & 1) & ~(Q & 1)); q[1] = q[0] = 4; return; }
  int C;
  if (((Q >> 1) & 3) == 3) { q[2] = 4; C = (((Q >> 3) & 3) << 3) | ((~(Q >> 5) & 3) << 1) | (Q & 1); }
  else { q[2] = (Q >> 5) & 3; C = Q & 31; }
  if ((C & 7) == 5) { q[1] = 4; q[0] = (C >> 3) & 3; } else { q[1] = (C >> 3) & 3; q[0] = C & 7; }
}
static int16_t g_trit_word[243]; static int16_t g_quint_word[125]; static int g_ise_ready;
static void ise_init(void) {
  if (g_ise_ready) return;
  for (int i = 0; i < 243; i++) g_trit_word[i] = -1;
  for (int i = 0; i < 125; i++) g_quint_word[i] = -1;
  for (int T = 0; T < 256; T++) { int t[5]; trits_of(T, t); const int k = t[0] + 3 * t[1] + 9 * t[2] + 27 * t[3] + 81 * t[4]; if (t[0] < 3 && t[1] < 3 && t[2] < 3 && g_trit_word[k] < 0) g_trit_word[k] = (int16_t)T; }
  for (int Q = 0; Q < 128; Q++) { int q[3]; quints_of(Q, q); const int k = q[0] + 5 * q[1] + 25 * q[2]; if (q[0] < 5 && q[1] < 5 && q[2] < 5 && g_quint_word[k] < 0) g_quint_word[k] = (int16_t)Q; }
  g_ise_ready = 1;
}
static int ise_bits(int n, int range) { const int b = R_BTQ[range][0]; return n * b + (R_BTQ[range][1] ? (8 * n + 4) / 5 : 0) + (R_BTQ[range][2] ? (7 * n + 2) / 3 : 0); }
static void ise_encode(uint8_t *blk, int *o, const uint8_t *v, int n, int range) {
  ise_init();
  const int bits = R_BTQ[range][0];
  if (R_BTQ[range][1]) {
    static const uint8_t sh[5] = { 0, 2, 4, 5, 7 }, nb[5] = { 2, 2, 1, 2, 1 };
    for (int g = 0; g < n; g += 5) {
      int k = 0, s = 1; for (int j = 0; j < 5; j++, s *= 3) k += (g + j < n ? v[g + j] >> bits : 0) * s;
      const int T = g_trit_word[k];
      for (int j = 0; j < 5 && g + j < n; j++) { put_bits(blk, o, v[g + j] & ((1u << bits) - 1u), bits); put_bits(blk, o, (uint32_t)(T >> sh[j]), nb[j]); }
    }
  } else if (R_BTQ[range][2]) {
    static const uint8_t sh[3] = { 0, 3, 5 }, nb[3] = { 3, 2, 2 };
    for (int g = 0; g < n; g += 3) {
      int k = 0, s = 1; for (int j = 0; j < 3; j++, s *= 5) k += (g + j < n ? v[g + j] >> bits : 0) * s;
      const int Q = g_quint_word[k];
      for (int j = 0; j < 3 && g + j < n; j++) { put_bits(blk, o, v[g + j] & ((1u << bits) - 1u), bits); put_bits(blk, o, (uint32_t)(Q >> sh[j]), nb[j]); }
    }
  } else for (int i = 0; i < n; i++) put_bits(blk, o, v[i], bits);
}

/* ---------------------------------------------------------------- UASTC -> ASTC 4x4 */
int uastc_to_astc(const uint8_t in[16], uint8_t out[16]) {
  uastc_lblock L; const int rc = uastc_unpack(in, &L); if (rc) return rc;
  memset(out, 0, 16);
  const int m = L.mode;
  if (m == 8) {                                         /* LDR void-extent block: 0x1FC, reserved bits 10-11 set, all-ones extent, 4 x UNORM16 */
    out[0] = 0xFC; out[1] = 0xFD; for (int i = 2; i < 8; i++) out[i] = 0xFF;
    for (int c = 0; c < 4; c++) { out[8 + 2 * c] = L.solid[c]; out[9 + 2 * c] = L.solid[c]; }
    return 0;
  }
  const int nc = UM_COMPS[m], range = UM_RANGE[m], planes = UM_PLANES[m], wb = UM_WBITS[m], maxw = (1 << wb) - 1;
  /* ASTC applies blue contraction when the sum of the second endpoint's RGB is smaller than the first's: swap the pair and
   * mirror every weight so that the block decodes to the same texels without it */
  int s0 = 0, s1 = 0;
  for (int c = 0; c < 3; c++) { s0 += astc_unquant_endpoint(range, L.ep[2 * c]); s1 += astc_unquant_endpoint(range, L.ep[2 * c + 1]); }
  if (s1 < s0) {
    for (int c = 0; c < nc; c++) { const uint8_t t = L.ep[2 * c]; L.ep[2 * c] = L.ep[2 * c + 1]; L.ep[2 * c + 1] = t; }
    for (int i = 0; i < 16 * planes; i++) L.w[i] = (uint8_t)(maxw - L.w[i]);
  }
  int o = 0;
  put_bits(out, &o, UM_ASTC_BM[m], 11); put_bits(out, &o, 0, 2); put_bits(out, &o, nc == 3 ? 8u : 12u, 4);
  ise_encode(out, &o, L.ep, 2 * nc, range);
  const int wtot = 16 * planes * wb;
  if (planes == 2) { int oc = 128 - wtot - 2; put_bits(out, &oc, (uint32_t)L.ccs, 2); }
  for (int i = 0; i < 16 * planes; i++) for (int b = 0; b < wb; b++) if ((L.w[i] >> b) & 1) { const int pos = 127 - (i * wb + b); out[pos >> 3] |= (uint8_t)(1u << (pos & 7)); }
  return 0;
}

/* ---------------------------------------------------------------- independent ASTC 4x4 decoder (single partition, LDR) */
static int ise_decode(const uint8_t *blk, int o, int rev, uint8_t *v, int n, int range) {
  /* rev: the sequence is stored from bit 127 downwards (weights) */
#define RD(nb) ({ uint32_t r_ = 0; for (int i_ = 0; i_ < (nb); i_++, o++) { const int p_ = rev ? 127 - o : o; r_ |= (uint32_t)((blk[p_ >> 3] >> (p_ & 7)) & 1) << i_; } r_; })
  const int bits = R_BTQ[range][0];
  if (R_BTQ[range][1]) {
    static const uint8_t sh[5] = { 0, 2, 4, 5, 7 }, nb[5] = { 2, 2, 1, 2, 1 };
    for (int g = 0; g < n; g += 5) {
      int T = 0, m[5] = { 0, 0, 0, 0, 0 };
      for (int j = 0; j < 5 && g + j < n; j++) { m[j] = (int)RD(bits); T |= (int)RD(nb[j]) << sh[j]; }
      int t[5]; trits_of(T, t);
      for (int j = 0; j < 5 && g + j < n; j++) v[g + j] = (uint8_t)((t[j] << bits) | m[j]);
    }
  } else if (R_BTQ[range][2]) {
    static const uint8_t sh[3] = { 0, 3, 5 }, nb[3] = { 3, 2, 2 };
    for (int g = 0; g < n; g += 3) {
      int Q = 0, m[3] = { 0, 0, 0 };
      for (int j = 0; j < 3 && g + j < n; j++) { m[j] = (int)RD(bits); Q |= (int)RD(nb[j]) << sh[j]; }
      int q[3]; quints_of(Q, q);
      for (int j = 0; j < 3 && g + j < n; j++) v[g + j] = (uint8_t)((q[j] << bits) | m[j]);
    }
  } else for (int i = 0; i < n; i++) v[i] = (uint8_t)RD(bits);
#undef RD
  return o;
}
static int astc_unquant_weight(int range, int v) {       /* general form (spec "Weight Unquantization") */
  const int bits = R_BTQ[range][0], tr = R_BTQ[range][1], qu = R_BTQ[range][2];
  if (!tr && !qu) return astc_unquant_weight_bits(bits, v);
  int r;
  if (bits == 0) { static const uint8_t t3[3] = { 0, 32, 63 }, q5[5] = { 0, 16, 32, 47, 63 }; r = tr ? t3[v] : q5[v]; }
  else {
    const int D = v >> bits, m = v & ((1 << bits) - 1), a = m & 1, b = (m >> 1) & 1, c = (m >> 2) & 1, A = a ? 0x7F : 0; int B = 0, C = 0;
    if (tr) { if (bits == 1) { C = 50; } else if (bits == 2) { C = 23; B = (b << 6) | (b << 2) | b; } else { C = 11; B = (c << 6) | (b << 5) | (c << 1) | b; } }
    else { if (bits == 1) { C = 28; } else { C = 13; B = (b << 6) | (b << 1); } }
    int T = D * C + B; T ^= A; r = (A & 0x20) | (T >> 2);
  }
  return r > 32 ? r + 1 : r;
}
int astc_decode_block(const uint8_t in[16], uint8_t rgba[64]) {
  const uint32_t lo = (uint32_t)in[0] | ((uint32_t)in[1] << 8);
  if ((lo & 0x1FF) == 0x1FC) {                          /* void extent */
    if (lo & 0x200) return -1;                          /* HDR */
    for (int i = 0; i < 16; i++) for (int c = 0; c < 4; c++) rgba[4 * i + c] = in[9 + 2 * c];       /* UNORM16 >> 8 */
    return 0;
  }
  const int bm = (int)(lo & 0x7FF);
  int W, H, R, Hb, D;
  if ((bm & 3) != 0) {                                  /* rows 1-5 of the 2D block-mode table */
    R = ((bm & 3) << 1) | ((bm >> 4) & 1); Hb = (bm >> 9) & 1; D = (bm >> 10) & 1;
    const int A = (bm >> 5) & 3, B = (bm >> 7) & 3;
    switch ((bm >> 2) & 3) {
      case 0: W = B + 4; H = A + 2; break;
      case 1: W = B + 8; H = A + 2; break;
      case 2: W = A + 2; H = B + 8; break;
      default: if (B & 2) { W = (B & 1) + 2; H = A + 2; } else { W = A + 2; H = (B & 1) + 6; } break;
    }
  } else return -2;                                     /* the other table rows never give a 4x4 grid */
  if (W != 4 || H != 4 || R < 2) return -2;
  static const int8_t wr_lo[8] = { -1, -1, 0, 1, 2, 3, 4, 5 }, wr_hi[8] = { -1, -1, 6, 7, 8, 9, 10, 11 };
  const int wrange = Hb ? wr_hi[R] : wr_lo[R];
  if (((in[1] >> 3) & 3) != 0) return -3;               /* partition count - 1: single partition only */
  const int cem = (int)(((uint32_t)in[1] >> 5) | (((uint32_t)in[2] & 1) << 3));
  if (cem != 8 && cem != 12) return -4;
  const int nw = D ? 32 : 16, wbits = ise_bits(nw, wrange), nv = cem == 8 ? 6 : 8;
  const int avail = 128 - 17 - wbits - (D ? 2 : 0);
  int crange = -1; for (int r = 20; r >= 0; r--) if (ise_bits(nv, r) <= avail) { crange = r; break; }
  if (crange < 4) return -5;                            /* the spec disallows fewer than 6 levels for colour values */
  uint8_t ev[8], wv[32];
  (void)ise_decode(in, 17, 0, ev, nv, crange);
  (void)ise_decode(in, 0, 1, wv, nw, wrange);
  int ccs = 0; if (D) { const int p = 128 - wbits - 2; ccs = ((in[p >> 3] >> (p & 7)) & 1) | (((in[(p + 1) >> 3] >> ((p + 1) & 7)) & 1) << 1); }
  int e0[4] = { 0, 0, 0, 255 }, e1[4] = { 0, 0, 0, 255 }, u[8];
  for (int i = 0; i < nv; i++) u[i] = astc_unquant_endpoint(crange, ev[i]);
  if (u[1] + u[3] + u[5] >= u[0] + u[2] + u[4]) { for (int c = 0; c < nv / 2; c++) { e0[c] = u[2 * c]; e1[c] = u[2 * c + 1]; } }
  else {                                                /* blue contraction + swap */
    const int a0 = nv == 8 ? u[7] : 255, a1 = nv == 8 ? u[6] : 255;
    e0[0] = (u[1] + u[5]) >> 1; e0[1] = (u[3] + u[5]) >> 1; e0[2] = u[5]; e0[3] = a0;
    e1[0] = (u[0] + u[4]) >> 1; e1[1] = (u[2] + u[4]) >> 1; e1[2] = u[4]; e1[3] = a1;
  }
  for (int i = 0; i < 16; i++) {
    const int w0 = astc_unquant_weight(wrange, wv[D ? 2 * i : i]), w1 = D ? astc_unquant_weight(wrange, wv[2 * i + 1]) : w0;
    for (int c = 0; c < 4; c++) rgba[4 * i + c] = (uint8_t)astc_interp(e0[c], e1[c], (D && c == ccs) ? w1 : w0);
  }
  return 0;
}

/* ---------------------------------------------------------------- encoder (deterministic, integer only) */
/* nearest representable endpoint value of a range: lowest unquantised value among equals, then the lowest code */
static uint8_t g_q_of[21][256]; static uint8_t g_uq[21][256]; static int g_q_ready;
static void quant_init(void) {
  if (g_q_ready) return;
  for (int r = 4; r < 21; r++) {
    for (int v = 0; v < R_LEVELS[r]; v++) g_uq[r][v] = (uint8_t)astc_unquant_endpoint(r, v);
    for (int x = 0; x < 256; x++) {
      int best = 0, bd = 1 << 30, bu = 0;
      for (int v = 0; v < R_LEVELS[r]; v++) { const int uqv = g_uq[r][v], d = uqv > x ? uqv - x : x - uqv; if (d < bd || (d == bd && uqv < bu)) { bd = d; best = v; bu = uqv; } }
      g_q_of[r][x] = (uint8_t)best;
    }
  }
  g_q_ready = 1;
}
static inline long long rdiv(long long n, long long d) { return n >= 0 ? (n + d / 2) / d : -((-n + d / 2) / d); }
static inline int bitlen64(unsigned long long v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

/* one plane: components comp[0..nc) of the 16 texels, endpoint range, weight bits -> endpoint codes (lo, hi), weights, SSE */
static uint32_t fit_plane(const uint8_t px[64], const int *comp, int nc, int range, int wb, uint8_t qlo[4], uint8_t qhi[4], uint8_t w[16]) {
  quant_init();
  const int nlev = 1 << wb;
  int lo[4], hi[4];
  if (nc == 1) {
    int mn = 255, mx = 0; for (int i = 0; i < 16; i++) { const int v = px[4 * i + comp[0]]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    lo[0] = mn; hi[0] = mx;
  } else {
    long long S[4] = { 0, 0, 0, 0 }, d[16][4], cov[4][4], v[4]; int mn[4], mx[4];
    for (int c = 0; c < nc; c++) { mn[c] = 255; mx[c] = 0; for (int i = 0; i < 16; i++) { const int x = px[4 * i + comp[c]]; S[c] += x; mn[c] = x < mn[c] ? x : mn[c]; mx[c] = x > mx[c] ? x : mx[c]; } }
    for (int i = 0; i < 16; i++) for (int c = 0; c < nc; c++) d[i][c] = 16 * (long long)px[4 * i + comp[c]] - S[c];
    for (int a = 0; a < nc; a++) for (int b = 0; b < nc; b++) { long long s = 0; for (int i = 0; i < 16; i++) s += d[i][a] * d[i][b]; cov[a][b] = s; }
    long long any = 0; for (int c = 0; c < nc; c++) { v[c] = mx[c] - mn[c]; any |= v[c]; }
    if (!any) for (int c = 0; c < nc; c++) v[c] = 1;
    for (int it = 0; it < 4; it++) {                    /* power iteration, renormalised to 15 bits by shifting */
      long long nv[4], m = 0;
      for (int a = 0; a < nc; a++) { long long s = 0; for (int b = 0; b < nc; b++) s += cov[a][b] * v[b]; nv[a] = s; const long long as = s < 0 ? -s : s; m = as > m ? as : m; }
      if (m == 0) break;
      const int sh = bitlen64((unsigned long long)m) - 15;
      for (int a = 0; a < nc; a++) v[a] = sh > 0 ? nv[a] / ((long long)1 << sh) : nv[a];
    }
    int ilo = 0, ihi = 0; long long plo = 0, phi = 0;
    for (int i = 0; i < 16; i++) { long long p = 0; for (int c = 0; c < nc; c++) p += d[i][c] * v[c]; if (i == 0 || p < plo) { plo = p; ilo = i; } if (i == 0 || p > phi) { phi = p; ihi = i; } }
    for (int c = 0; c < nc; c++) { lo[c] = px[4 * ilo + comp[c]]; hi[c] = px[4 * ihi + comp[c]]; }
  }
  uint32_t best_sse = 0xffffffffu;
  for (int pass = 0; pass < 2; pass++) {
    uint8_t ql[4], qh[4], ww[16]; int pal[32][4];
    for (int c = 0; c < nc; c++) { ql[c] = g_q_of[range][lo[c]]; qh[c] = g_q_of[range][hi[c]]; }
    for (int k = 0; k < nlev; k++) { const int uw = astc_unquant_weight_bits(wb, k); for (int c = 0; c < nc; c++) pal[k][c] = astc_interp(g_uq[range][ql[c]], g_uq[range][qh[c]], uw); }
    /* weight of a texel: the level its projection on the endpoint line rounds to, and that level's two neighbours, compared under
     * the exact interpolation (lowest level among equals) — 3 instead of up to 32 evaluations per texel */
    int den = 0, dl[4];
    for (int c = 0; c < nc; c++) { dl[c] = (int)g_uq[range][qh[c]] - (int)g_uq[range][ql[c]]; den += dl[c] * dl[c]; }
    uint32_t sse = 0;
    for (int i = 0; i < 16; i++) {
      int num = 0; for (int c = 0; c < nc; c++) num += ((int)px[4 * i + comp[c]] - (int)g_uq[range][ql[c]]) * dl[c];
      int k0 = 0; if (den > 0) { const int t = num < 0 ? 0 : (num > den ? den : num); k0 = (t * (nlev - 1) + den / 2) / den; }
      uint32_t be = 0xffffffffu; int bk = 0;
      for (int k = k0 - 1; k <= k0 + 1; k++) { if (k < 0 || k >= nlev) continue; uint32_t e = 0; for (int c = 0; c < nc; c++) { const int dd = pal[k][c] - (int)px[4 * i + comp[c]]; e += (uint32_t)(dd * dd); } if (e < be) { be = e; bk = k; } }
      ww[i] = (uint8_t)bk; sse += be;
    }
    if (sse < best_sse) { best_sse = sse; memcpy(qlo, ql, 4); memcpy(qhi, qh, 4); memcpy(w, ww, 16); }
    if (pass == 1) break;
    /* least-squares endpoints for the weights just found (exact integer normal equations) */
    long long Suu = 0, Svv = 0, Suv = 0;
    for (int i = 0; i < 16; i++) { const long long u = astc_unquant_weight_bits(wb, ww[i]), vv = 64 - u; Suu += u * u; Svv += vv * vv; Suv += u * vv; }
    const long long det = Svv * Suu - Suv * Suv;
    if (det <= 0) break;
    for (int c = 0; c < nc; c++) {
      long long Suc = 0, Svc = 0;
      for (int i = 0; i < 16; i++) { const long long u = astc_unquant_weight_bits(wb, ww[i]), x = px[4 * i + comp[c]]; Suc += u * x; Svc += (64 - u) * x; }
      long long a = rdiv(64 * (Suu * Svc - Suv * Suc), det), b = rdiv(64 * (Svv * Suc - Suv * Svc), det);
      lo[c] = (int)(a < 0 ? 0 : (a > 255 ? 255 : a)); hi[c] = (int)(b < 0 ? 0 : (b > 255 ? 255 : b));
    }
  }
  return best_sse;
}

/* ETC1 hint of a half block (2 x 4 columns x0..x0+1 of the decoded texels): best intensity table under a 4-bit base colour */
static int etc1_inten_hint(const uint8_t dec[64], int x0) {
  static const int lo_[8] = { 2, 5, 9, 13, 18, 24, 33, 47 }, hi_[8] = { 8, 17, 29, 42, 60, 80, 106, 183 };
  int base[3];
  for (int c = 0; c < 3; c++) { int s = 0; for (int y = 0; y < 4; y++) for (int x = x0; x < x0 + 2; x++) s += dec[4 * (4 * y + x) + c]; const int avg = (s + 4) / 8; base[c] = ((avg * 15 + 127) / 255) * 17; }
  int bt = 0; uint32_t be = 0xffffffffu;
  for (int t = 0; t < 8; t++) {
    uint32_t e = 0; const int mod[4] = { -hi_[t], -lo_[t], lo_[t], hi_[t] };
    for (int y = 0; y < 4; y++) for (int x = x0; x < x0 + 2; x++) {
      uint32_t bs = 0xffffffffu;
      for (int s = 0; s < 4; s++) { uint32_t es = 0; for (int c = 0; c < 3; c++) { int v = base[c] + mod[s]; v = v < 0 ? 0 : (v > 255 ? 255 : v); const int dd = v - (int)dec[4 * (4 * y + x) + c]; es += (uint32_t)(dd * dd); } bs = es < bs ? es : bs; }
      e += bs;
    }
    if (e < be) { be = e; bt = t; }
  }
  return bt;
}

void uastc_encode_lblock(const uint8_t px[64], uastc_lblock *out) {
  memset(out, 0, sizeof(*out));
  int same = 1, alpha = 0;
  for (int i = 0; i < 16; i++) { if (memcmp(px + 4 * i, px, 4)) same = 0; if (px[4 * i + 3] != 255) alpha = 1; }
  if (same) {
    out->mode = 8; memcpy(out->solid, px, 4);
    /* ETC1 hint of a solid block: differential mode, the (table, selector, 5-bit base) triple closest to the colour */
    static const int lo_[8] = { 2, 5, 9, 13, 18, 24, 33, 47 }, hi_[8] = { 8, 17, 29, 42, 60, 80, 106, 183 };
    uint32_t be = 0xffffffffu;
    for (int t = 0; t < 8; t++) for (int s = 0; s < 4; s++) {
      const int mod = s == 0 ? -hi_[t] : (s == 1 ? -lo_[t] : (s == 2 ? lo_[t] : hi_[t]));
      uint32_t e = 0; uint8_t b5[3];
      for (int c = 0; c < 3; c++) { uint32_t bc = 0xffffffffu; int bb = 0; for (int q = 0; q < 32; q++) { int v = ((q << 3) | (q >> 2)) + mod; v = v < 0 ? 0 : (v > 255 ? 255 : v); const int dd = v - (int)px[c]; if ((uint32_t)(dd * dd) < bc) { bc = (uint32_t)(dd * dd); bb = q; } } e += bc; b5[c] = (uint8_t)bb; }
      if (e < be) { be = e; out->etc1_diff = 1; out->etc1_inten0 = t; out->etc1_sel = s; memcpy(out->etc1_base, b5, 3); }
    }
    return;
  }
  /* candidates in preference order; the first with the lowest SSE wins */
  uastc_lblock first_rgb; memset(&first_rgb, 0, sizeof first_rgb);
  /* opaque blocks: mode 0, mode 18, and the dual-plane mode 6 with its second plane on the channel the mode-0 fit serves worst
   * (largest squared error; the lowest channel among equals); alpha blocks: modes 10, 12 and 11 (second plane = alpha) */
  int cand_rgb[3][2] = { { 0, -1 }, { 18, -1 }, { 6, 0 } }; static const int cand_a[3][2] = { { 10, -1 }, { 12, -1 }, { 11, 3 } };
  const int ncand = 3; uint32_t best = 0xffffffffu;
  for (int k = 0; k < ncand; k++) {
    if (!alpha && k == 2) {
      uint8_t d0[64]; uastc_lblock_rgba(&first_rgb, d0);
      uint32_t ce[3] = { 0, 0, 0 };
      for (int i = 0; i < 16; i++) for (int c = 0; c < 3; c++) { const int dd = (int)d0[4 * i + c] - (int)px[4 * i + c]; ce[c] += (uint32_t)(dd * dd); }
      cand_rgb[2][1] = ce[1] > ce[0] ? (ce[2] > ce[1] ? 2 : 1) : (ce[2] > ce[0] ? 2 : 0);
    }
    const int m = alpha ? cand_a[k][0] : cand_rgb[k][0], ccs = alpha ? cand_a[k][1] : cand_rgb[k][1];
    const int nc = UM_COMPS[m], range = UM_RANGE[m], wb = UM_WBITS[m];
    uastc_lblock L; memset(&L, 0, sizeof L); L.mode = m; L.ccs = ccs < 0 ? 0 : ccs;
    uint32_t sse;
    if (ccs < 0) {
      const int comp[4] = { 0, 1, 2, 3 }; uint8_t ql[4], qh[4], w[16];
      sse = fit_plane(px, comp, nc, range, wb, ql, qh, w);
      for (int c = 0; c < nc; c++) { L.ep[2 * c] = ql[c]; L.ep[2 * c + 1] = qh[c]; }
      memcpy(L.w, w, 16);
    } else {
      int comp0[4], n0 = 0; for (int c = 0; c < nc; c++) if (c != ccs) comp0[n0++] = c;
      const int comp1[1] = { ccs }; uint8_t ql[4], qh[4], w0[16], q1l[4], q1h[4], w1[16];
      sse = fit_plane(px, comp0, n0, range, wb, ql, qh, w0);
      sse += fit_plane(px, comp1, 1, range, wb, q1l, q1h, w1);
      for (int j = 0; j < n0; j++) { L.ep[2 * comp0[j]] = ql[j]; L.ep[2 * comp0[j] + 1] = qh[j]; }
      L.ep[2 * ccs] = q1l[0]; L.ep[2 * ccs + 1] = q1h[0];
      for (int i = 0; i < 16; i++) { L.w[2 * i] = w0[i]; L.w[2 * i + 1] = w1[i]; }
    }
    if (!alpha && k == 0) first_rgb = L;
    if (sse < best) { best = sse; *out = L; }
  }
  /* anchor rule: the first weight of every plane is stored without its top bit -> mirror the plane when that bit is set
   * (a plane's endpoints are the components it interpolates, so the two planes of a dual-plane block mirror independently) */
  { const int m = out->mode, nc = UM_COMPS[m], planes = UM_PLANES[m], wb = UM_WBITS[m], maxw = (1 << wb) - 1;
    for (int p = 0; p < planes; p++) if (out->w[p] > maxw / 2) {
      for (int c = 0; c < nc; c++) if (planes == 1 || (p == 1) == (c == out->ccs)) { const uint8_t t = out->ep[2 * c]; out->ep[2 * c] = out->ep[2 * c + 1]; out->ep[2 * c + 1] = t; }
      for (int i = 0; i < 16; i++) out->w[planes * i + p] = (uint8_t)(maxw - out->w[planes * i + p]);
    } }
  /* transcoder hints: ETC1 halves are the left / right 2x4 columns (flip 0), individual 4-bit base colours (diff 0), no bias;
   * BC1 hints 0 (the transcoder then fits BC1 endpoints itself); ETC2 alpha hint: table 13, multiplier from the alpha span */
  uint8_t dec[64]; uastc_lblock_rgba(out, dec);
  out->etc1_flip = 0; out->etc1_diff = 0; out->etc1_bias = 0; out->bc1_hint0 = out->bc1_hint1 = 0;
  out->etc1_inten0 = etc1_inten_hint(dec, 0); out->etc1_inten1 = etc1_inten_hint(dec, 2);
  if (UM_ALPHA[out->mode]) { int mn = 255, mx = 0; for (int i = 0; i < 16; i++) { const int a = dec[4 * i + 3]; mn = a < mn ? a : mn; mx = a > mx ? a : mx; } int mul = (mx - mn + 19) / 20; mul = mul < 1 ? 1 : (mul > 15 ? 15 : mul); out->etc2_hints = (mul << 4) | 13; }
}
void uastc_encode_block(const uint8_t px[64], uint8_t out[16]) { uastc_lblock L; uastc_encode_lblock(px, &L); uastc_pack(&L, out); }

/* ---------------------------------------------------------------- images and the KTX2 container */
static void fetch_block(const uint8_t *img, uint32_t W, uint32_t H, int yflip, uint32_t bx, uint32_t by, uint8_t px[64]) {
  for (int y = 0; y < 4; y++) {
    uint32_t py = by * 4 + (uint32_t)y; if (py >= H) py = H - 1;
    const uint32_t sr = yflip ? H - 1 - py : py;
    for (int x = 0; x < 4; x++) { uint32_t pxx = bx * 4 + (uint32_t)x; if (pxx >= W) pxx = W - 1; memcpy(px + 4 * (4 * y + x), img + 4 * ((size_t)sr * W + pxx), 4); }
  }
}
int uastc_ktx2_encode(const uint8_t *const *layers, int n_layers, uint32_t W, uint32_t H, int y_flip, orc_buf *out) {
  if (n_layers < 1 || !W || !H) return -1;
  const uint32_t bx = (W + 3) / 4, by = (H + 3) / 4; const size_t nb = (size_t)bx * by;
  int any_alpha = 0;
  for (int l = 0; l < n_layers && !any_alpha; l++) for (size_t i = 0; i < (size_t)W * H; i++) if (layers[l][4 * i + 3] != 255) { any_alpha = 1; break; }
  static const uint8_t ident[12] = { 0xAB, 'K', 'T', 'X', ' ', '2', '0', 0xBB, '\r', '\n', 0x1A, '\n' };
  static const char writer[] = "uvol-mi355x uastc 0.1";
  orc_buf kvd = { 0 };
  { ob_u32(&kvd, 12 + 12); ob_bytes(&kvd, "KTXanimData", 12); ob_u32(&kvd, 1); ob_u32(&kvd, 15); ob_u32(&kvd, 0);
    const uint32_t wl = 10 + (uint32_t)sizeof(writer); ob_u32(&kvd, wl); ob_bytes(&kvd, "KTXwriter", 10); ob_bytes(&kvd, writer, sizeof(writer)); while (kvd.n & 3) ob_u8(&kvd, 0); }
  const uint32_t dfd_off = 80 + 24, dfd_len = 44, kvd_off = dfd_off + dfd_len, kvd_len = (uint32_t)kvd.n;
  const uint64_t lvl_off = ((uint64_t)kvd_off + kvd_len + 15) & ~15ull, lvl_len = (uint64_t)n_layers * nb * 16;
  ob_bytes(out, ident, 12);
  ob_u32(out, 0); ob_u32(out, 1); ob_u32(out, W); ob_u32(out, H); ob_u32(out, 0); ob_u32(out, (uint32_t)n_layers); ob_u32(out, 1); ob_u32(out, 1); ob_u32(out, 0);
  ob_u32(out, dfd_off); ob_u32(out, dfd_len); ob_u32(out, kvd_off); ob_u32(out, kvd_len); ob_u64(out, 0); ob_u64(out, 0);
  ob_u64(out, lvl_off); ob_u64(out, lvl_len); ob_u64(out, lvl_len);
  /* DFD: colour model 166 (UASTC), BT.709 primaries, sRGB transfer, 4x4 texel blocks of 16 bytes, one 128-bit sample */
  ob_u32(out, 44); ob_u32(out, 0); ob_u16(out, 2); ob_u16(out, 40);
  ob_u8(out, 166); ob_u8(out, 1); ob_u8(out, 2); ob_u8(out, 0);
  ob_u8(out, 3); ob_u8(out, 3); ob_u8(out, 0); ob_u8(out, 0);
  ob_u8(out, 16); for (int i = 0; i < 7; i++) ob_u8(out, 0);
  ob_u16(out, 0); ob_u8(out, 127); ob_u8(out, any_alpha ? 3 : 0); ob_u8(out, 0); ob_u8(out, 0); ob_u8(out, 0); ob_u8(out, 0); ob_u32(out, 0); ob_u32(out, 0xFFFFFFFFu);
  ob_bytes(out, kvd.p, kvd.n);
  while (out->n < lvl_off) ob_u8(out, 0);
  for (int l = 0; l < n_layers; l++) for (uint32_t y = 0; y < by; y++) for (uint32_t x = 0; x < bx; x++) {
    uint8_t px[64], blk[16]; fetch_block(layers[l], W, H, y_flip, x, y, px); uastc_encode_block(px, blk); ob_bytes(out, blk, 16);
  }
  ob_free(&kvd);
  return 0;
}
int uastc_ktx2_info(const uint8_t *b, size_t n, uint32_t *W, uint32_t *H, uint32_t *layers, uint64_t *lvl_off, int *has_alpha) {
  static const uint8_t ident[12] = { 0xAB, 'K', 'T', 'X', ' ', '2', '0', 0xBB, '\r', '\n', 0x1A, '\n' };
  if (!b || n < 104 + 44 || memcmp(b, ident, 12)) return -1;
  uint32_t u[9]; memcpy(u, b + 12, 36);
  if (u[0] != 0 || u[8] != 0 || u[7] != 1 || u[6] != 1 || !u[2] || !u[3]) return -2;
  uint32_t dfd_off, dfd_len; memcpy(&dfd_off, b + 48, 4); memcpy(&dfd_len, b + 52, 4);
  if (dfd_len < 44 || dfd_off > n || dfd_len > n - dfd_off || b[dfd_off + 12] != 166) return -3;
  uint64_t lo, ll; memcpy(&lo, b + 80, 8); memcpy(&ll, b + 88, 8);
  const uint32_t L = u[5] ? u[5] : 1; const uint64_t need = (uint64_t)L * ((u[2] + 3) / 4) * ((u[3] + 3) / 4) * 16;
  if (lo > n || ll > n - lo || ll != need) return -4;
  *W = u[2]; *H = u[3]; *layers = L; *lvl_off = lo; if (has_alpha) *has_alpha = (b[dfd_off + 31] & 15) == 3;
  return 0;
}
/* target 0: RGBA8 layers (W*H*4 each, stored row order); target 1: ASTC 4x4 blocks (bx*by*16 per layer) */
/* ---- UASTC -> BC7 (target 2): what the stock loader asks a UASTC source for on every desktop GPU (reference src/lib/KTX2Loader.js:601-609
 * BC7_M5 for UASTC, chosen at :665-676 when ASTC is not supported).  The basis transcoder's own tables are not in the reference; this is a
 * deterministic re-fit of the block's OWN endpoints and texels, gated by PSNR against the RGBA32 decode (tests), not by bit parity with stock:
 *  - single-plane modes (0, 10, 12, 18) and solid blocks -> BC7 mode 6 (one subset, RGBA 7-bit endpoints + a p-bit each, 4-bit indices):
 *    each endpoint takes the p-bit under which its four 8-bit values are represented best, every texel the index of the nearest of the
 *    16 interpolated colours;
 *  - dual-plane modes (6, 11) -> BC7 mode 5 (RGB 7-bit endpoints with 2-bit indices, a separate 8-bit scalar channel with its own 2-bit
 *    indices; the ROTATION puts the block's second-plane channel there): UASTC's and BC7's 2-bit weights are the same {0, 21, 43, 64}.
 * Bit layouts as in tex_decode.hip (LSB first); pixel 0's index has its top bit implied 0 (else endpoints swapped, indices complemented). */
static const int BC7_W2[4] = { 0, 21, 43, 64 };
static const int BC7_W4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };
static void bc7_put(uint64_t w[2], int *pos, uint64_t v, int nbits) {
  if (*pos < 64) { w[0] |= v << *pos; if (*pos + nbits > 64) w[1] |= v >> (64 - *pos); } else w[1] |= v << (*pos - 64);
  *pos += nbits;
}
int uastc_to_bc7(const uint8_t in[16], uint8_t out[16]) {
  uastc_lblock L; const int rc = uastc_unpack(in, &L); if (rc) return rc;
  uint8_t px[64]; uastc_lblock_rgba(&L, px);
  int lo[4] = { 0, 0, 0, 255 }, hi[4] = { 0, 0, 0, 255 }, dual = 0, ccs = 0;
  if (L.mode == 8) { for (int c = 0; c < 4; c++) lo[c] = hi[c] = L.solid[c]; }
  else {
    const int range = UM_RANGE[L.mode], nc = UM_COMPS[L.mode];
    for (int c = 0; c < nc; c++) { lo[c] = astc_unquant_endpoint(range, L.ep[2 * c]); hi[c] = astc_unquant_endpoint(range, L.ep[2 * c + 1]); }
    dual = UM_PLANES[L.mode] == 2; ccs = L.ccs;
  }
  uint64_t w[2] = { 0, 0 }; int pos;
  if (!dual) {
    int e7[2][4], pb[2] = { 0, 0 };
    for (int s = 0; s < 2; s++) {
      const int *v = s ? hi : lo; int best = 1 << 30;
      for (int p = 0; p < 2; p++) {
        int q[4], err = 0;
        for (int c = 0; c < 4; c++) { int t = (v[c] - p + 1) >> 1; t = t < 0 ? 0 : (t > 127 ? 127 : t); q[c] = t; const int d = ((t << 1) | p) - v[c]; err += d * d; }
        if (err < best) { best = err; pb[s] = p; for (int c = 0; c < 4; c++) e7[s][c] = q[c]; }
      }
    }
    int idx[16];
    for (int i = 0; i < 16; i++) {
      int bw = 0, be = 1 << 30;
      for (int k = 0; k < 16; k++) {
        int err = 0;
        for (int c = 0; c < 4; c++) { const int a = (e7[0][c] << 1) | pb[0], b = (e7[1][c] << 1) | pb[1], d = ((a * (64 - BC7_W4[k]) + b * BC7_W4[k] + 32) >> 6) - px[4 * i + c]; err += d * d; }
        if (err < be) { be = err; bw = k; }
      }
      idx[i] = bw;
    }
    const int swap = idx[0] >= 8;
    w[0] = 1ull << 6; pos = 7;
    for (int c = 0; c < 4; c++) { bc7_put(w, &pos, (uint64_t)e7[swap ? 1 : 0][c], 7); bc7_put(w, &pos, (uint64_t)e7[swap ? 0 : 1][c], 7); }
    bc7_put(w, &pos, (uint64_t)pb[swap ? 1 : 0], 1); bc7_put(w, &pos, (uint64_t)pb[swap ? 0 : 1], 1);
    for (int i = 0; i < 16; i++) bc7_put(w, &pos, (uint64_t)(swap ? 15 - idx[i] : idx[i]), i == 0 ? 3 : 4);
  } else {
    const int rot = ccs == 3 ? 0 : ccs + 1;
    int src[3]; for (int k = 0; k < 3; k++) src[k] = (ccs < 3 && k == ccs) ? 3 : k;       /* encoded colour channel k holds this actual channel */
    int e7[2][3];
    for (int s = 0; s < 2; s++) for (int k = 0; k < 3; k++) {
      const int v = (s ? hi : lo)[src[k]]; int bt = 0, bd = 1 << 30;
      for (int t = (v >> 1) - 1; t <= (v >> 1) + 1; t++) { if (t < 0 || t > 127) continue; int d = ((t << 1) | (t >> 6)) - v; d = d < 0 ? -d : d; if (d < bd) { bd = d; bt = t; } }
      e7[s][k] = bt;
    }
    const int a0 = lo[ccs], a1 = hi[ccs];
    int ci[16], ai[16];
    for (int i = 0; i < 16; i++) {
      int bw = 0, be = 1 << 30;
      for (int k = 0; k < 4; k++) {
        int err = 0;
        for (int c = 0; c < 3; c++) { const int a = (e7[0][c] << 1) | (e7[0][c] >> 6), b = (e7[1][c] << 1) | (e7[1][c] >> 6), d = ((a * (64 - BC7_W2[k]) + b * BC7_W2[k] + 32) >> 6) - px[4 * i + src[c]]; err += d * d; }
        if (err < be) { be = err; bw = k; }
      }
      ci[i] = bw; bw = 0; be = 1 << 30;
      for (int k = 0; k < 4; k++) { const int d = ((a0 * (64 - BC7_W2[k]) + a1 * BC7_W2[k] + 32) >> 6) - px[4 * i + ccs]; if (d * d < be) { be = d * d; bw = k; } }
      ai[i] = bw;
    }
    const int cswap = ci[0] >= 2, aswap = ai[0] >= 2;
    w[0] = 1ull << 5; pos = 6;
    bc7_put(w, &pos, (uint64_t)rot, 2);
    for (int c = 0; c < 3; c++) { bc7_put(w, &pos, (uint64_t)e7[cswap ? 1 : 0][c], 7); bc7_put(w, &pos, (uint64_t)e7[cswap ? 0 : 1][c], 7); }
    bc7_put(w, &pos, (uint64_t)(aswap ? a1 : a0), 8); bc7_put(w, &pos, (uint64_t)(aswap ? a0 : a1), 8);
    for (int i = 0; i < 16; i++) bc7_put(w, &pos, (uint64_t)(cswap ? 3 - ci[i] : ci[i]), i == 0 ? 1 : 2);
    for (int i = 0; i < 16; i++) bc7_put(w, &pos, (uint64_t)(aswap ? 3 - ai[i] : ai[i]), i == 0 ? 1 : 2);
  }
  memcpy(out, &w[0], 8); memcpy(out + 8, &w[1], 8);
  return 0;
}
int uastc_ktx2_decode(const uint8_t *b, size_t n, int target, uint8_t *out) {
  uint32_t W, H, L; uint64_t lo;
  const int rc = uastc_ktx2_info(b, n, &W, &H, &L, &lo, NULL); if (rc) return rc;
  const uint32_t bx = (W + 3) / 4, by = (H + 3) / 4;
  for (uint32_t l = 0; l < L; l++) for (uint32_t y = 0; y < by; y++) for (uint32_t x = 0; x < bx; x++) {
    const uint8_t *blk = b + lo + 16 * (((size_t)l * by + y) * bx + x);
    if (target == 1) { if (uastc_to_astc(blk, out + 16 * (((size_t)l * by + y) * bx + x))) return -10; continue; }
    if (target == 2) { if (uastc_to_bc7(blk, out + 16 * (((size_t)l * by + y) * bx + x))) return -10; continue; }
    uint8_t px[64]; if (uastc_decode_block(blk, px)) return -10;
    for (int yy = 0; yy < 4 && 4 * y + (uint32_t)yy < H; yy++) for (int xx = 0; xx < 4 && 4 * x + (uint32_t)xx < W; xx++)
      memcpy(out + 4 * (((size_t)l * H + 4 * y + (uint32_t)yy) * W + 4 * x + (uint32_t)xx), px + 4 * (4 * yy + xx), 4);
  }
  return 0;
}
