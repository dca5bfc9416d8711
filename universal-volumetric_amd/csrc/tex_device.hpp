// tex_device.hpp — device-side data model + helpers of the texture (.ktx2 / ETC1S / BasisLZ) pipeline.
//
// Replaces the arithmetic of one `basisu -ktx2 -tex_type video -multifile_num B -y_flip` process
// (scripts/Encoder.py:290) for one segment of B layers resident in HBM.  Bitstream: SURVEY.md
// Appendix B (KTX2 container B.0, Huffman tables B.1, codebooks B.2, slices B.3, block model B.4).
//
// HBM layout: source layers are RGBA8 row-major (16 B = one 4-texel block row per lane, so a wave
// reads 1 KiB contiguous per block row); every per-block array is a flat SoA array of L*bx*by
// entries (layer-major, raster order inside a layer) so parallel kernels coalesce.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#define TEX_MAX_LAYERS 64
#define TEX_MAX_CODEBOOK 16128
#define TEX_HS 64                       // selector history size
#define TEX_VQ_ROUNDS 24
#define TEX_NMODEL 9                    // 0 endpoint_pred, 1 delta_endpoint, 2 selector, 3 selector_rle, 4..6 colour5 delta, 7 inten delta, 8 selector byte delta
#define TEX_MODEL_CAP (TEX_MAX_CODEBOOK + TEX_HS + 8)
#define TEX_E_ALPHA (-60)               // device status of the opaque layout: a texel with alpha != 255 (the host encodes the segment again with alpha slices)

struct TexVQ {
  uint32_t n_items, K, nl, done, m_round, round_active;
  uint32_t *leaf;                          // [n_items]
  unsigned long long *stW, *stS, *stQ;     // [K], [K*DIM], [K*DIM]
  uint8_t *splittable, *chosen; int32_t *axis; long long *th, *prio; uint32_t *newidx;   // [K]
  uint32_t *split;                         // [K] selector VQ: this round's split of a leaf in one word (chosen << 31 | axis << 24 | threshold << 16 | new leaf), 0 = not split
};

struct TexHuff { uint32_t n; uint32_t *freq; uint8_t *size; uint16_t *code; };   // code bit-reversed (LSB-first emission)

struct TexJob {
  const uint8_t *layer[TEX_MAX_LAYERS];
  uint32_t W, H, L, bx, by, nb, NB;          // L = SLICES: images << ashift (with alpha every image has a colour and an alpha slice: rgb0 a0 rgb1 a1 ...)
  uint32_t ashift;                            // 1: alpha slices; slice v shows image v >> 1, kind v & 1, and follows slice v - 2 of its kind
  int32_t yflip; uint32_t Kmax_e, Kmax_s, T_skip;
  int32_t status;
  uint8_t *skip;            // [NB]
  uint32_t *cell;           // [NB] per-block (t<<15|r<<10|g<<5|b)
  uint32_t *hist;           // [1<<18]
  uint8_t *flag; uint32_t *bsum;   // scan scratch
  uint32_t ncell; uint32_t *cid, *cw, *cidx;
  TexVQ vq[2];              // 0 endpoints (dim 4, items = cells), 1 selectors (dim 16, items = coded blocks)
  uint32_t *ent;            // [Ke] entry tuple per cluster
  uint32_t *ecb; uint32_t ne; uint32_t *emap;   // unique sorted endpoint codebook, cluster -> index
  uint16_t *bei, *bsi; uint32_t *bsel;          // [NB]
  uint32_t n_items; uint32_t *item;             // coded (non-skipped) blocks, ascending
  uint32_t *scb; uint8_t *sused; uint32_t *scu; uint32_t ns; uint32_t *smap;
  uint8_t *pred;            // [NB]
  uint32_t *clist; uint32_t ncoded[TEX_MAX_LAYERS];   // coded (non-skipped) blocks per slice, raster order
  uint8_t *msym; uint32_t *gid, *gstart, *gsize;       // macroblock symbols and their equal-value groups
  unsigned long long *tok;  // [NB*3]  kind | sym<<8 | extra<<32
  TexHuff hm[TEX_NMODEL];
  uint32_t *hscratch;       // Huffman build scratch
  uint8_t *sec[3]; uint32_t sec_cap[3], sec_len[3];     // endpoints / selectors / tables bit sections
  uint8_t *slice[TEX_MAX_LAYERS]; uint32_t slice_cap, slice_len[TEX_MAX_LAYERS];
  unsigned long long pack_off, pack_len;                 // this segment's sections + slices, concatenated, in the batch's packed output buffer
  unsigned long long slice_bits[TEX_MAX_LAYERS];
  uint32_t n_skipped;
};

__device__ __forceinline__ int t_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int t_expand5(int c) { return (c << 3) | (c >> 2); }
__device__ __forceinline__ int t_inten(int t, int s) {
  // g_etc1_inten_tables, linear selector order low -> high
  const int hi_[8] = { 8, 17, 29, 42, 60, 80, 106, 183 }, lo_[8] = { 2, 5, 9, 13, 18, 24, 33, 47 };
  const int m = (s == 0 || s == 3) ? hi_[t] : lo_[t];
  return s < 2 ? -m : m;
}

// 4x4 block fetch with y-flip and edge replication; px[i] = R | G<<8 | B<<16 (i = y*4+x)
// amask (optional): AND of the texels' alpha bytes (bits 24..31), for the opacity check of the ETC1S path
// With alpha slices (J.ashift) `l` is a slice: odd slices are the image's alpha channel as the grey image (a, a, a).
__device__ inline void t_load_block(const TexJob &J, uint32_t l, uint32_t X, uint32_t Y, uint32_t px[16], uint32_t *amask = nullptr) {
  const uint8_t *img = J.layer[l >> J.ashift];
  const bool akind = (l & J.ashift) != 0;
  for (int y = 0; y < 4; y++) {
    uint32_t py = Y * 4 + y; if (py >= J.H) py = J.H - 1;
    const uint32_t sr = J.yflip ? J.H - 1 - py : py;
    const uint8_t *row = img + 4 * ((size_t)sr * J.W);
    if (X * 4 + 3 < J.W && (J.W & 3) == 0) {
      const uint4 v = *reinterpret_cast<const uint4 *>(row + 16 * (size_t)X);
      if (akind) { px[4 * y + 0] = (v.x >> 24) * 0x010101u; px[4 * y + 1] = (v.y >> 24) * 0x010101u; px[4 * y + 2] = (v.z >> 24) * 0x010101u; px[4 * y + 3] = (v.w >> 24) * 0x010101u; }
      else { px[4 * y + 0] = v.x & 0xffffffu; px[4 * y + 1] = v.y & 0xffffffu; px[4 * y + 2] = v.z & 0xffffffu; px[4 * y + 3] = v.w & 0xffffffu; }
      if (amask) *amask &= v.x & v.y & v.z & v.w;
    } else {
      for (int x = 0; x < 4; x++) { uint32_t pxx = X * 4 + x; if (pxx >= J.W) pxx = J.W - 1; const uint8_t *p = row + 4 * (size_t)pxx; px[4 * y + x] = akind ? (uint32_t)p[3] * 0x010101u : ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)); if (amask) *amask &= (uint32_t)p[3] << 24; }
    }
  }
}

// SSE of a block under (c5, t) with per-texel optimal selectors; optionally the selector word and the
// per-texel error rows (4 x u16 packed in a u64, selector s at bits 16s) used by the codebook search.
template <bool WANT_SEL, bool WANT_ROWS>
__device__ inline uint32_t t_eval_block(const uint32_t px[16], int c5r, int c5g, int c5b, int t, uint32_t *sel_out, unsigned long long *rows) {
  const int br = t_expand5(c5r), bg = t_expand5(c5g), bb = t_expand5(c5b);
  int cr[4], cg[4], cb[4];
  for (int s = 0; s < 4; s++) { const int d = t_inten(t, s); cr[s] = t_clampi(br + d, 0, 255); cg[s] = t_clampi(bg + d, 0, 255); cb[s] = t_clampi(bb + d, 0, 255); }
  uint32_t tot = 0, sel = 0;
  if (!WANT_ROWS) {
    // No channel clamps under any of the four modifiers (the usual case away from black / white): the error of selector s is
    // A + 2 d_s B + 3 d_s^2 with A = sum_c (base_c - p_c)^2, B = sum_c (base_c - p_c) — the same integers as the per-channel
    // form below, for 17 instead of 36 operations per texel.
    const int d0 = t_inten(t, 0), d1 = t_inten(t, 1), d2 = t_inten(t, 2), d3 = t_inten(t, 3);
    const int lo = br < bg ? (br < bb ? br : bb) : (bg < bb ? bg : bb), hi = br > bg ? (br > bb ? br : bb) : (bg > bb ? bg : bb);
    if (!WANT_SEL && lo + d0 >= 0 && hi + d3 <= 255) {
      // totals only (k_tex_fit; round 6): the tables are symmetric (d = -hi, -lo, lo, hi), so the best selector's error is
      // A + min(3 hi^2 - hi |B2|, 3 lo^2 - lo |B2|) - the same integer as the four-way minimum below, without the selector bookkeeping
      const int kh = 3 * d3 * d3, kl = 3 * d2 * d2;
      for (int i = 0; i < 16; i++) {
        const int er = br - (int)(px[i] & 255), eg = bg - (int)((px[i] >> 8) & 255), eb = bb - (int)((px[i] >> 16) & 255);
        const int A = er * er + eg * eg + eb * eb, B2 = 2 * (er + eg + eb), ab = B2 < 0 ? -B2 : B2;
        const int a = kh - d3 * ab, b = kl - d2 * ab;
        tot += (uint32_t)(A + (a < b ? a : b));
      }
      return tot;
    }
    if (lo + d0 >= 0 && hi + d3 <= 255) {
      const int k0 = 3 * d0 * d0, k1 = 3 * d1 * d1, k2 = 3 * d2 * d2, k3 = 3 * d3 * d3;
      for (int i = 0; i < 16; i++) {
        const int er = br - (int)(px[i] & 255), eg = bg - (int)((px[i] >> 8) & 255), eb = bb - (int)((px[i] >> 16) & 255);
        const int A = er * er + eg * eg + eb * eb, B2 = 2 * (er + eg + eb);
        const uint32_t e0 = (uint32_t)(A + d0 * B2 + k0), e1 = (uint32_t)(A + d1 * B2 + k1), e2 = (uint32_t)(A + d2 * B2 + k2), e3 = (uint32_t)(A + d3 * B2 + k3);
        uint32_t be = e0; int bs = 0;
        if (e1 < be) { be = e1; bs = 1; }
        if (e2 < be) { be = e2; bs = 2; }
        if (e3 < be) { be = e3; bs = 3; }
        tot += be; sel |= (uint32_t)bs << (2 * i);
      }
      if (WANT_SEL) *sel_out = sel;
      return tot;
    }
  }
  for (int i = 0; i < 16; i++) {
    const int r = (int)(px[i] & 255), g = (int)((px[i] >> 8) & 255), b = (int)((px[i] >> 16) & 255);
    uint32_t be = 0xffffffffu; int bs = 0; unsigned long long row = 0;
    for (int s = 0; s < 4; s++) {
      const int dr = cr[s] - r, dg = cg[s] - g, db = cb[s] - b;
      const uint32_t e = (uint32_t)(dr * dr + dg * dg + db * db);
      if (WANT_ROWS) row |= (unsigned long long)(e > 65535u ? 65535u : e) << (16 * s);
      if (e < be) { be = e; bs = s; }
    }
    tot += be; sel |= (uint32_t)bs << (2 * i);
    if (WANT_ROWS) rows[i] = row;
  }
  if (WANT_SEL) *sel_out = sel;
  return tot;
}

// One 16x16x64 signed-i8 matrix-core tile (v_mfma_i32_16x16x64_i8): acc[r] += sum_k A[row][k] * B[k][col] with
// row = 4 * (lane >> 4) + r, col = lane & 15.  Lane l supplies 16 k-values of A's row (l & 15) and 16 k-values of B's column
// (l & 15), both for k-group (l >> 4); A and B use the same lane/byte -> k map, so a caller that builds both operands itself
// only depends on the row / column / accumulator maps stated here.
#ifdef HIPEMU
__device__ inline void t_mfma_i8_16x16x64(const uint32_t a[4], const uint32_t b[4], int acc[4]) {
  const int lane = hipemu_lane(), col = lane & 15, rg = lane >> 4;
  const unsigned long long a01 = a[0] | ((unsigned long long)a[1] << 32), a23 = a[2] | ((unsigned long long)a[3] << 32);
  const unsigned long long b01 = b[0] | ((unsigned long long)b[1] << 32), b23 = b[2] | ((unsigned long long)b[3] << 32);
  bool ok;
  for (int g = 0; g < 4; g++) {
    const unsigned long long B0 = hipemu_wave_exchange(b01, col + 16 * g, &ok), B1 = hipemu_wave_exchange(b23, col + 16 * g, &ok);
    for (int r = 0; r < 4; r++) {
      const unsigned long long A0 = hipemu_wave_exchange(a01, 4 * rg + r + 16 * g, &ok), A1 = hipemu_wave_exchange(a23, 4 * rg + r + 16 * g, &ok);
      int sum = 0;
      for (int j = 0; j < 8; j++) sum += (int)(int8_t)(A0 >> (8 * j)) * (int)(int8_t)(B0 >> (8 * j)) + (int)(int8_t)(A1 >> (8 * j)) * (int)(int8_t)(B1 >> (8 * j));
      acc[r] += sum;
    }
  }
}
#else
typedef int t_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void t_mfma_i8_16x16x64(const uint32_t a[4], const uint32_t b[4], int acc[4]) {
  const t_v4i va = { (int)a[0], (int)a[1], (int)a[2], (int)a[3] }, vb = { (int)b[0], (int)b[1], (int)b[2], (int)b[3] }, vc = { acc[0], acc[1], acc[2], acc[3] };
  const t_v4i vd = __builtin_amdgcn_mfma_i32_16x16x64_i8(va, vb, vc, 0, 0, 0);
  acc[0] = vd.x; acc[1] = vd.y; acc[2] = vd.z; acc[3] = vd.w;
}
#endif

__device__ __forceinline__ void t_cell_coords(uint32_t cell, int x[4]) {
  x[0] = t_expand5((cell >> 10) & 31); x[1] = t_expand5((cell >> 5) & 31); x[2] = t_expand5(cell & 31); x[3] = t_inten((cell >> 15) & 7, 3);
}

// ---- device bit writer (single thread), LSB first ----
struct TBitW { uint8_t *p; uint32_t cap, n; unsigned long long acc; int nacc; int overflow; };
__device__ inline void tb_init(TBitW &w, uint8_t *p, uint32_t cap) { w.p = p; w.cap = cap; w.n = 0; w.acc = 0; w.nacc = 0; w.overflow = 0; }
__device__ inline void tb_put(TBitW &w, uint32_t v, int n) {
  while (n > 0) {
    const int k = n > 24 ? 24 : n;
    w.acc |= (unsigned long long)(v & ((1u << k) - 1)) << w.nacc; w.nacc += k; v >>= k; n -= k;
    while (w.nacc >= 8) { if (w.n < w.cap) w.p[w.n] = (uint8_t)(w.acc & 0xff); else w.overflow = 1; w.n++; w.acc >>= 8; w.nacc -= 8; }
  }
}
__device__ inline void tb_flush(TBitW &w) { if (w.nacc > 0) { if (w.n < w.cap) w.p[w.n] = (uint8_t)(w.acc & 0xff); else w.overflow = 1; w.n++; w.acc = 0; w.nacc = 0; } }

// vlc as (bits, len): chunks of cb bits, low chunk first, each followed by a continuation flag
__device__ __forceinline__ void t_vlc(uint32_t v, int cb, unsigned long long &bits, int &len) {
  bits = 0; len = 0;
  for (;;) { const uint32_t chunk = v & ((1u << cb) - 1); v >>= cb; bits |= (unsigned long long)(chunk | (v ? (1u << cb) : 0u)) << len; len += cb + 1; if (!v) break; }
}
