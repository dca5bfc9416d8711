// tex_uastc.hip — hand-written HIP (gfx950) UASTC LDR 4x4 texture mode: RGBA8 layers -> UASTC blocks -> KTX2, and the decode
// side UASTC -> RGBA8 / ASTC 4x4.
//
// What it stands for in the reference: `basisu -uastc -ktx2 -tex_type video ...` — the second texture mode of the encoder the
// reference driver spawns (scripts/Encoder.py:290 passes no -uastc; the north star and SURVEY §8 f4 ask for the mode) — and,
// on the consumer side, the UASTC -> ASTC 4x4 transcode the stock player's KTX2Loader requests first for UASTC sources
// (src/lib/KTX2Loader.js:591-600, :648-689).  Formats: UASTC LDR 4x4 specification (basis_universal) and the ASTC LDR profile;
// the single-subset modes 0, 6, 18 (opaque), 10, 11, 12 (alpha) and 8 (solid) are emitted — DESIGN.md §2 says what pins them
// (parity with basisu itself is unpinned: no binary, no fixture).
//
// One thread per 4x4 block, blocks are independent: a streaming kernel (64 B of texels in, 16 B out per block) whose
// arithmetic is a handful of exact integer least-squares fits per block.  Everything is integer and deterministic (the tests compare
// every output byte with the CPU restatement of the same algorithm).  No MFMA: there is no contraction; the bound is VALU integer throughput and, for the transcodes, HBM.
#include "uvol_common.hpp"
#include <dlfcn.h>
#include <algorithm>

// ---- mode tables (UASTC specification) ----
struct UMode { uint8_t huff, hufflen, wbits, range, comps, planes, bias, bc1h1, alpha; uint16_t astc_bm; };
__device__ __forceinline__ UMode u_mode(int m) {
  // only the emitted modes; {prefix code, length, weight bits, endpoint range, components, planes, has bias, has bc1 hint 1, has alpha, ASTC block mode}
  switch (m) {
    case 0: return UMode{0x1, 4, 4, 19, 3, 1, 1, 1, 0, 0x242};
    case 6: return UMode{0x1B, 5, 2, 18, 3, 2, 1, 1, 0, 0x442};
    case 10: return UMode{0x2, 3, 4, 13, 4, 1, 0, 0, 1, 0x242};
    case 11: return UMode{0x0, 2, 2, 13, 4, 2, 0, 0, 1, 0x442};
    case 12: return UMode{0x6, 3, 3, 19, 4, 1, 0, 0, 1, 0x53};
    default: return UMode{0x9, 4, 5, 11, 3, 1, 1, 1, 0, 0x253};     // 18
  }
}
// ASTC integer-sequence parameters of the endpoint ranges in use: 11 = 32 levels, 13 = 48, 18 = 160, 19 = 192
__device__ __forceinline__ int u_rbits(int range) { return range == 11 ? 5 : (range == 13 ? 4 : (range == 18 ? 5 : 6)); }
__device__ __forceinline__ int u_rtrit(int range) { return (range == 13 || range == 19) ? 1 : 0; }
__device__ __forceinline__ int u_rquint(int range) { return range == 18 ? 1 : 0; }
__device__ __forceinline__ int u_rlevels(int range) { return range == 11 ? 32 : (range == 13 ? 48 : (range == 18 ? 160 : 192)); }
__device__ __forceinline__ int u_slot(int range) { return range == 11 ? 0 : (range == 13 ? 1 : (range == 18 ? 2 : 3)); }

// endpoint unquantisation tables of those four ranges and their inverses (nearest level), built on the host at context creation
struct UTab { uint8_t uq[4][256]; uint8_t qof[4][256]; };

static int u_host_unquant(int bits, int tr, int qu, int v) {          // ASTC "endpoint unquantization"
  if (!tr && !qu) { int r = 0, have = 0; while (have < 8) { r = (r << bits) | v; have += bits; } return (r >> (have - 8)) & 255; }
  const int D = v >> bits, m = v & ((1 << bits) - 1), a = m & 1, b = (m >> 1) & 1, c = (m >> 2) & 1, d = (m >> 3) & 1, e = (m >> 4) & 1, f = (m >> 5) & 1, A = a ? 0x1FF : 0;
  int B = 0, C = 0;
  if (tr) { if (bits == 4) { C = 22; B = (d << 8) | (c << 7) | (b << 6) | (d << 2) | (c << 1) | b; } else { C = 5; B = (f << 8) | (e << 7) | (d << 6) | (c << 5) | (b << 4) | f; } }
  else { C = 6; B = (e << 8) | (d << 7) | (c << 6) | (b << 5) | e; }
  int T = D * C + B; T ^= A;
  return (A & 0x80) | (T >> 2);
}
static void u_host_tables(UTab &T) {
  static const int ranges[4] = { 11, 13, 18, 19 }, bits[4] = { 5, 4, 5, 6 }, tr[4] = { 0, 1, 0, 1 }, qu[4] = { 0, 0, 1, 0 }, lev[4] = { 32, 48, 160, 192 };
  (void)ranges;
  memset(&T, 0, sizeof T);
  for (int s = 0; s < 4; s++) {
    for (int v = 0; v < lev[s]; v++) T.uq[s][v] = (uint8_t)u_host_unquant(bits[s], tr[s], qu[s], v);
    for (int x = 0; x < 256; x++) {                                   // nearest level; among equals the lower value, then the lower code
      int best = 0, bd = 1 << 30, bu = 0;
      for (int v = 0; v < lev[s]; v++) { const int u = T.uq[s][v], dd = u > x ? u - x : x - u; if (dd < bd || (dd == bd && u < bu)) { bd = dd; best = v; bu = u; } }
      T.qof[s][x] = (uint8_t)best;
    }
  }
}

__device__ __forceinline__ int u_wunq(int wb, int k) {                 // weight index -> 0..64 (bit replication, > 32 moves up by one)
  const int r = wb == 2 ? 21 * k : (wb == 3 ? 9 * k : (wb == 4 ? ((k << 2) | (k >> 2)) : (wb == 5 ? ((k << 1) | (k >> 4)) : 63 * k)));
  return r > 32 ? r + 1 : r;
}
__device__ __forceinline__ int u_interp(int l, int h, int w) { l = (l << 8) | l; h = (h << 8) | h; return ((l * (64 - w) + h * w + 32) >> 6) >> 8; }
// round-to-nearest division (halves away from zero), 0 < d < 2^33, |n| < 2^41: the quotient through the f64 divider (both operands
// are exact in a double, so it is off by at most one) + an exact integer correction, instead of the ~300-instruction 64-bit
// integer division sequence - the least-squares refits divide 6 times per plane fit, uniformly across the 16 lanes of a block
__device__ __forceinline__ long long u_udiv_exact(long long nn, long long d) {          // floor(nn / d), nn >= 0
#ifdef HIPEMU
  return nn / d;
#else
  long long q = (long long)((double)nn / (double)d);
  const long long r = nn - q * d;
  if (r < 0) q--; else if (r >= d) q++;
  return q;
#endif
}
__device__ __forceinline__ long long u_rdiv(long long n, long long d) { return n >= 0 ? u_udiv_exact(n + d / 2, d) : -u_udiv_exact(-n + d / 2, d); }
__device__ __forceinline__ int u_comp(uint32_t px, int c) { return (int)((px >> (8 * c)) & 255u); }

// 128-bit block under construction (LSB first)
struct UBits {
  unsigned long long lo, hi;
  __device__ __forceinline__ void put(int &o, uint32_t v, int n) {
    if (n <= 0) return;
    const unsigned long long x = (unsigned long long)v & ((1ull << n) - 1ull);
    if (o < 64) { lo |= x << o; if (o + n > 64) hi |= x >> (64 - o); } else hi |= x << (o - 64);
    o += n;
  }
  __device__ __forceinline__ uint32_t get(int &o, int n) const {
    if (n <= 0) return 0;
    unsigned long long x;
    if (o < 64) { x = lo >> o; if (o + n > 64) x |= hi << (64 - o); } else x = hi >> (o - 64);
    o += n;
    return (uint32_t)(x & ((1ull << n) - 1ull));
  }
};

// logical block: mode, second-plane component, endpoint codes (ASTC order c0.lo c0.hi c1.lo ...), weights (planes interleaved)
struct ULog { int mode, ccs; uint8_t ep[8]; uint8_t w[32]; };

__device__ inline void u_decode_log(const ULog &L, const UTab *T, uint32_t out[16]) {
  const UMode M = u_mode(L.mode); const int slot = u_slot(M.range);
  int lo[4] = { 0, 0, 0, 255 }, hi[4] = { 0, 0, 0, 255 };
  for (int c = 0; c < M.comps; c++) { lo[c] = T->uq[slot][L.ep[2 * c]]; hi[c] = T->uq[slot][L.ep[2 * c + 1]]; }
  for (int i = 0; i < 16; i++) {
    const int w0 = u_wunq(M.wbits, L.w[M.planes * i]), w1 = M.planes == 2 ? u_wunq(M.wbits, L.w[2 * i + 1]) : w0;
    uint32_t v = 0;
    for (int c = 0; c < 4; c++) v |= (uint32_t)(c < M.comps ? u_interp(lo[c], hi[c], (M.planes == 2 && c == L.ccs) ? w1 : w0) : 255) << (8 * c);
    out[i] = v;
  }
}
// ------------------------------------------------------------------------------------------------
// encoder: one lane per 4x4 block.  Per plane: principal axis by an integer power iteration on the 16x-scaled covariance,
// endpoints = the texels at the ends of the axis, per texel the weight its projection on the endpoint line rounds to or one of
// that level's two neighbours (exact ASTC interpolation decides), then ONE exact integer least-squares refit of the endpoints
// for those weights, kept when it lowers the error.  Opaque blocks try mode 0, mode 18
// and the dual-plane mode 6 with its second plane on the channel mode 0 serves worst; alpha blocks modes 10, 12, 11.
// (A 16-lanes-per-block variant with DPP reductions was measured at a third of this kernel's rate: most of the work per block
// - axis, quantisation, refit, bit packing - is uniform across a block's texels and was replicated 16 times.)
// ------------------------------------------------------------------------------------------------
// weights of one plane: 8 bits per texel in two words (texel i: bits 8 (i & 7) of w[i >> 3])
struct UW16 { unsigned long long w[2]; };
__device__ __forceinline__ int uw_get(const UW16 &W, int i) { return (int)((W.w[i >> 3] >> (8 * (i & 7))) & 255u); }

__device__ inline uint32_t u_fit_plane(const uint32_t px[16], int cmask, int range, int wb, const UTab *T, uint8_t qlo[4], uint8_t qhi[4], UW16 &W) {
  const int slot = u_slot(range), nlev = 1 << wb;
  int comp[4] = { 0, 0, 0, 0 }, nc = 0;
  for (int c = 0; c < 4; c++) if ((cmask >> c) & 1) comp[nc++] = c;
  int lo[4] = { 0, 0, 0, 0 }, hi[4] = { 0, 0, 0, 0 };
  if (nc == 1) {
    int mn = 255, mx = 0;
    for (int i = 0; i < 16; i++) { const int v = u_comp(px[i], comp[0]); mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    lo[0] = mn; hi[0] = mx;
  } else {
    // 32-bit where the ranges allow it: |16 c - S| <= 4080, a covariance entry < 2^29, the axis is renormalised below 2^15, a
    // projection is below 2^31: the same values a 64-bit evaluation gives
    int S[4] = { 0, 0, 0, 0 }, cov[4][4], mn[4] = { 255, 255, 255, 255 }, mx[4] = { 0, 0, 0, 0 }; long long v[4] = { 0, 0, 0, 0 };
    for (int c = 0; c < nc; c++) for (int i = 0; i < 16; i++) { const int x = u_comp(px[i], comp[c]); S[c] += x; mn[c] = x < mn[c] ? x : mn[c]; mx[c] = x > mx[c] ? x : mx[c]; }
    for (int a = 0; a < nc; a++) for (int b = a; b < nc; b++) { int sv = 0; for (int i = 0; i < 16; i++) sv += (16 * u_comp(px[i], comp[a]) - S[a]) * (16 * u_comp(px[i], comp[b]) - S[b]); cov[a][b] = sv; cov[b][a] = sv; }
    long long any = 0; for (int c = 0; c < nc; c++) { v[c] = mx[c] - mn[c]; any |= v[c]; }
    if (!any) for (int c = 0; c < nc; c++) v[c] = 1;
    for (int it = 0; it < 4; it++) {
      long long nv[4] = { 0, 0, 0, 0 }, m = 0;
      for (int a = 0; a < nc; a++) { long long sv = 0; for (int b = 0; b < nc; b++) sv += (long long)cov[a][b] * (int)v[b]; nv[a] = sv; const long long as = sv < 0 ? -sv : sv; m = as > m ? as : m; }
      if (m == 0) break;
      const int sh = (64 - __clzll(m)) - 15;
      for (int a = 0; a < nc; a++) v[a] = sh > 0 ? ((nv[a] + ((nv[a] >> 63) & (((long long)1 << sh) - 1))) >> sh) : nv[a];      // truncating division by 2^sh
    }
    int ilo = 0, ihi = 0; long long plo = 0, phi = 0;
    for (int i = 0; i < 16; i++) {
      long long p = 0; for (int c = 0; c < nc; c++) p += (long long)(16 * u_comp(px[i], comp[c]) - S[c]) * (int)v[c];
      if (i == 0 || p < plo) { plo = p; ilo = i; }
      if (i == 0 || p > phi) { phi = p; ihi = i; }
    }
    for (int c = 0; c < nc; c++) { lo[c] = u_comp(px[ilo], comp[c]); hi[c] = u_comp(px[ihi], comp[c]); }
  }
  uint32_t best_sse = 0xffffffffu;
  for (int pass = 0; pass < 2; pass++) {
    uint8_t ql[4] = { 0, 0, 0, 0 }, qh[4] = { 0, 0, 0, 0 }; int ul[4] = { 0, 0, 0, 0 }, uh[4] = { 0, 0, 0, 0 }, dl[4] = { 0, 0, 0, 0 }, den = 0;
    for (int c = 0; c < nc; c++) { ql[c] = T->qof[slot][lo[c]]; qh[c] = T->qof[slot][hi[c]]; ul[c] = T->uq[slot][ql[c]]; uh[c] = T->uq[slot][qh[c]]; dl[c] = uh[c] - ul[c]; den += dl[c] * dl[c]; }
    UW16 ww; ww.w[0] = 0; ww.w[1] = 0; uint32_t sse = 0;
    int Suu = 0, Svv = 0, Suv = 0, Suc[4] = { 0, 0, 0, 0 }, Svc[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 16; i++) {
      int num = 0; for (int c = 0; c < nc; c++) num += (u_comp(px[i], comp[c]) - ul[c]) * dl[c];
      int k0 = 0; if (den > 0) { const int t = num < 0 ? 0 : (num > den ? den : num); k0 = (int)u_udiv_exact((long long)(t * (nlev - 1) + den / 2), (long long)den); }
      uint32_t be = 0xffffffffu; int bk = 0;
      for (int k = k0 - 1; k <= k0 + 1; k++) {
        if (k < 0 || k >= nlev) continue;
        const int uw = u_wunq(wb, k); uint32_t e = 0;
        for (int c = 0; c < nc; c++) { const int dd = u_interp(ul[c], uh[c], uw) - u_comp(px[i], comp[c]); e += (uint32_t)(dd * dd); }
        if (e < be) { be = e; bk = k; }
      }
      ww.w[i >> 3] |= (unsigned long long)bk << (8 * (i & 7)); sse += be;
      const int u = u_wunq(wb, bk), vv = 64 - u;                         // sums of the least-squares refit, gathered on the way
      Suu += u * u; Svv += vv * vv; Suv += u * vv;
      for (int c = 0; c < nc; c++) { const int x = u_comp(px[i], comp[c]); Suc[c] += u * x; Svc[c] += vv * x; }
    }
    if (sse < best_sse) { best_sse = sse; for (int c = 0; c < 4; c++) { qlo[c] = ql[c]; qhi[c] = qh[c]; } W = ww; }
    if (pass == 1) break;
    const long long det = (long long)Svv * Suu - (long long)Suv * Suv;
    if (det <= 0) break;
    for (int c = 0; c < nc; c++) {
      const long long a = u_rdiv(64 * ((long long)Suu * Svc[c] - (long long)Suv * Suc[c]), det), b = u_rdiv(64 * ((long long)Svv * Suc[c] - (long long)Suv * Svc[c]), det);
      lo[c] = (int)(a < 0 ? 0 : (a > 255 ? 255 : a)); hi[c] = (int)(b < 0 ? 0 : (b > 255 ? 255 : b));
    }
  }
  return best_sse;
}

// 16 texels (px[i] = R | G << 8 | B << 16 | A << 24, i = 4 y + x) -> one UASTC block
__device__ inline void u_encode_block(const uint32_t px[16], const UTab *T, UBits &B) {
  B.lo = 0; B.hi = 0; int o = 0;
  bool same = true, alpha = false;
  for (int i = 0; i < 16; i++) { same &= px[i] == px[0]; alpha |= (px[i] >> 24) != 255u; }
  const int lo_[8] = { 2, 5, 9, 13, 18, 24, 33, 47 }, hi_[8] = { 8, 17, 29, 42, 60, 80, 106, 183 };
  if (same) {                                                           // mode 8: the colour + the ETC1 hint (table, selector, 5-bit base) closest to it
    uint32_t be = 0xffffffffu; int bt = 0, bsel = 0, b5[3] = { 0, 0, 0 };
    for (int t = 0; t < 8; t++) for (int sl = 0; sl < 4; sl++) {
      const int mod = sl == 0 ? -hi_[t] : (sl == 1 ? -lo_[t] : (sl == 2 ? lo_[t] : hi_[t]));
      uint32_t e = 0; int cb[3];
      for (int c = 0; c < 3; c++) { uint32_t bc = 0xffffffffu; int bb = 0; for (int q = 0; q < 32; q++) { int v = ((q << 3) | (q >> 2)) + mod; v = v < 0 ? 0 : (v > 255 ? 255 : v); const int dd = v - u_comp(px[0], c); if ((uint32_t)(dd * dd) < bc) { bc = (uint32_t)(dd * dd); bb = q; } } e += bc; cb[c] = bb; }
      if (e < be) { be = e; bt = t; bsel = sl; b5[0] = cb[0]; b5[1] = cb[1]; b5[2] = cb[2]; }
    }
    B.put(o, 0x17, 5);
    for (int c = 0; c < 4; c++) B.put(o, (uint32_t)u_comp(px[0], c), 8);
    B.put(o, 1, 1); B.put(o, (uint32_t)bt, 3); B.put(o, (uint32_t)bsel, 2);
    for (int c = 0; c < 3; c++) B.put(o, (uint32_t)b5[c], 5);
    return;
  }
  const int cand_a[3][2] = { { 10, -1 }, { 12, -1 }, { 11, 3 } };
  uint32_t best = 0xffffffffu; int ccs6 = 0;
  int Rmode = 0, Rccs = 0; uint8_t Rep[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; UW16 R0, R1; R0.w[0] = R0.w[1] = R1.w[0] = R1.w[1] = 0;
  for (int k = 0; k < 3; k++) {
    const int m = alpha ? cand_a[k][0] : (k == 0 ? 0 : (k == 1 ? 18 : 6)), ccs = alpha ? cand_a[k][1] : (k == 2 ? ccs6 : -1);
    const UMode M = u_mode(m); const int all = (1 << M.comps) - 1;
    uint8_t ep[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; UW16 w0, w1; w0.w[0] = w0.w[1] = w1.w[0] = w1.w[1] = 0; uint32_t sse;
    if (ccs < 0) {
      uint8_t ql[4], qh[4];
      sse = u_fit_plane(px, all, M.range, M.wbits, T, ql, qh, w0);
      for (int c = 0; c < M.comps; c++) { ep[2 * c] = ql[c]; ep[2 * c + 1] = qh[c]; }
      if (!alpha && k == 0) {                                           // per-channel error of the mode-0 fit -> second-plane channel of mode 6
        const int sl = u_slot(M.range); uint32_t ce[3] = { 0, 0, 0 };
        for (int i = 0; i < 16; i++) { const int uw = u_wunq(M.wbits, uw_get(w0, i)); for (int c = 0; c < 3; c++) { const int dd = u_interp(T->uq[sl][ql[c]], T->uq[sl][qh[c]], uw) - u_comp(px[i], c); ce[c] += (uint32_t)(dd * dd); } }
        ccs6 = ce[1] > ce[0] ? (ce[2] > ce[1] ? 2 : 1) : (ce[2] > ce[0] ? 2 : 0);
      }
    } else {
      uint8_t ql[4], qh[4], q1l[4], q1h[4];
      sse = u_fit_plane(px, all & ~(1 << ccs), M.range, M.wbits, T, ql, qh, w0);
      sse += u_fit_plane(px, 1 << ccs, M.range, M.wbits, T, q1l, q1h, w1);
      int j = 0;
      for (int c = 0; c < M.comps; c++) if (c != ccs) { ep[2 * c] = ql[j]; ep[2 * c + 1] = qh[j]; j++; }
      ep[2 * ccs] = q1l[0]; ep[2 * ccs + 1] = q1h[0];
    }
    if (sse < best) { best = sse; Rmode = m; Rccs = ccs < 0 ? 0 : ccs; R0 = w0; R1 = w1; for (int i = 0; i < 8; i++) Rep[i] = ep[i]; }
  }
  const UMode M = u_mode(Rmode); const int maxw = (1 << M.wbits) - 1;
  for (int p = 0; p < M.planes; p++) {                                  // anchor rule: a plane whose first weight has its top bit set is mirrored
    UW16 &Wp = p == 0 ? R0 : R1;
    if (uw_get(Wp, 0) > maxw / 2) {
      for (int c = 0; c < M.comps; c++) if (M.planes == 1 || (p == 1) == (c == Rccs)) { const uint8_t t = Rep[2 * c]; Rep[2 * c] = Rep[2 * c + 1]; Rep[2 * c + 1] = t; }
      const unsigned long long mx8 = 0x0101010101010101ull * (unsigned long long)maxw;
      Wp.w[0] = mx8 - Wp.w[0]; Wp.w[1] = mx8 - Wp.w[1];                  // every byte <= maxw: no borrow between bytes
    }
  }
  // decoded texels -> transcoder hints: ETC1 halves are the left / right 2x4 columns (flip 0), individual 4-bit base colours (diff 0),
  // no bias; BC1 hints 0; ETC2 alpha hint: table 13, multiplier from the alpha span
  const int slot = u_slot(M.range); uint32_t dec[16];
  for (int i = 0; i < 16; i++) {
    const int uw0 = u_wunq(M.wbits, uw_get(R0, i)), uw1 = M.planes == 2 ? u_wunq(M.wbits, uw_get(R1, i)) : uw0; uint32_t v = 0;
    for (int c = 0; c < 4; c++) v |= (uint32_t)(c < M.comps ? u_interp(T->uq[slot][Rep[2 * c]], T->uq[slot][Rep[2 * c + 1]], (M.planes == 2 && c == Rccs) ? uw1 : uw0) : 255) << (8 * c);
    dec[i] = v;
  }
  int inten[2] = { 0, 0 };
  for (int h = 0; h < 2; h++) {
    const int x0 = 2 * h; int base[3];
    for (int c = 0; c < 3; c++) { int sm = 0; for (int y = 0; y < 4; y++) for (int x = x0; x < x0 + 2; x++) sm += u_comp(dec[4 * y + x], c); const int avg = (sm + 4) / 8; base[c] = ((avg * 15 + 127) / 255) * 17; }
    uint32_t be = 0xffffffffu;
    for (int t = 0; t < 8; t++) {
      uint32_t e = 0; const int mod[4] = { -hi_[t], -lo_[t], lo_[t], hi_[t] };
      for (int y = 0; y < 4; y++) for (int x = x0; x < x0 + 2; x++) {
        uint32_t bs = 0xffffffffu;
        for (int sl = 0; sl < 4; sl++) { uint32_t es = 0; for (int c = 0; c < 3; c++) { int v = base[c] + mod[sl]; v = v < 0 ? 0 : (v > 255 ? 255 : v); const int dd = v - u_comp(dec[4 * y + x], c); es += (uint32_t)(dd * dd); } bs = es < bs ? es : bs; }
        e += bs;
      }
      if (e < be) { be = e; inten[h] = t; }
    }
  }
  int etc2 = 0;
  if (M.alpha) { int mn = 255, mx = 0; for (int i = 0; i < 16; i++) { const int a = (int)(dec[i] >> 24); mn = a < mn ? a : mn; mx = a > mx ? a : mx; } int mul = (mx - mn + 19) / 20; mul = mul < 1 ? 1 : (mul > 15 ? 15 : mul); etc2 = (mul << 4) | 13; }
  B.put(o, M.huff, M.hufflen);
  B.put(o, 0, 1); if (M.bc1h1) B.put(o, 0, 1);
  B.put(o, 0, 1); B.put(o, 0, 1); B.put(o, (uint32_t)inten[0], 3); B.put(o, (uint32_t)inten[1], 3);
  if (M.bias) B.put(o, 0, 5);
  if (M.alpha) B.put(o, (uint32_t)etc2, 8);
  if (M.planes == 2) B.put(o, (uint32_t)Rccs, 2);
  const int nv = 2 * M.comps, bits = u_rbits(M.range), tr = u_rtrit(M.range), qu = u_rquint(M.range);
  if (tr || qu) {
    const int per = tr ? 5 : 3, mul = tr ? 3 : 5;
    for (int g = 0; g * per < nv; g++) {
      uint32_t acc = 0, scale = 1; int cnt = 0;
      for (int k = 0; k < per && g * per + k < nv; k++, cnt++) { acc += (uint32_t)(Rep[g * per + k] >> bits) * scale; scale *= (uint32_t)mul; }
      const int nb = cnt == per ? (tr ? 8 : 7) : (tr ? (cnt == 1 ? 2 : (cnt == 2 ? 4 : (cnt == 3 ? 5 : 7))) : (cnt == 1 ? 3 : 5));
      B.put(o, acc, nb);
    }
  }
  for (int i = 0; i < nv; i++) B.put(o, (uint32_t)Rep[i] & ((1u << bits) - 1u), bits);
  for (int i = 0; i < 16; i++) {
    B.put(o, (uint32_t)uw_get(R0, i), i == 0 ? M.wbits - 1 : M.wbits);
    if (M.planes == 2) B.put(o, (uint32_t)uw_get(R1, i), i == 0 ? M.wbits - 1 : M.wbits);
  }
}

// ------------------------------------------------------------------------------------------------
// decode side: block -> logical block
// ------------------------------------------------------------------------------------------------
// returns 0, or a negative code for a mode this codec does not emit / a corrupt block; solid blocks: mode 8, colour in `solid`
__device__ inline int u_unpack(const UBits &B, ULog &L, uint32_t &solid) {
  int o = 0; int m = -1;
  { const uint32_t b7 = (uint32_t)(B.lo & 127u);
    if ((b7 & 15u) == 0x1) { m = 0; o = 4; } else if ((b7 & 15u) == 0x9) { m = 18; o = 4; } else if ((b7 & 31u) == 0x1B) { m = 6; o = 5; } else if ((b7 & 31u) == 0x17) { m = 8; o = 5; }
    else if ((b7 & 7u) == 0x2) { m = 10; o = 3; } else if ((b7 & 3u) == 0x0) { m = 11; o = 2; } else if ((b7 & 7u) == 0x6) { m = 12; o = 3; } }
  if (m < 0) return -2;
  L.mode = m; L.ccs = 0;
  if (m == 8) { solid = B.get(o, 8); solid |= B.get(o, 8) << 8; solid |= B.get(o, 8) << 16; solid |= B.get(o, 8) << 24; return 0; }
  const UMode M = u_mode(m);
  o += 1 + (M.bc1h1 ? 1 : 0) + 8 + (M.bias ? 5 : 0) + (M.alpha ? 8 : 0);           // hints are for other transcode targets
  if (M.planes == 2) L.ccs = (int)B.get(o, 2);
  const int nv = 2 * M.comps, bits = u_rbits(M.range), tr = u_rtrit(M.range), qu = u_rquint(M.range), per = tr ? 5 : 3, mul = tr ? 3 : 5;
  uint32_t tq[4] = { 0, 0, 0, 0 }; int groups = 0;
  if (tr || qu) for (int g = 0; g * per < nv; g++, groups++) {
    const int cnt = nv - g * per < per ? nv - g * per : per;
    const int nb = cnt == per ? (tr ? 8 : 7) : (tr ? (cnt == 1 ? 2 : (cnt == 2 ? 4 : (cnt == 3 ? 5 : 7))) : (cnt == 1 ? 3 : 5));
    tq[g] = B.get(o, nb);
  }
  for (int i = 0; i < nv; i++) {
    uint32_t v = B.get(o, bits);
    if (groups) { uint32_t a = tq[i / per]; for (int k = 0; k < i % per; k++) a /= (uint32_t)mul; v |= (a % (uint32_t)mul) << bits; }
    if ((int)v >= u_rlevels(M.range)) return -3;
    L.ep[i] = (uint8_t)v;
  }
  for (int i = 0; i < 16 * M.planes; i++) L.w[i] = (uint8_t)B.get(o, i < M.planes ? M.wbits - 1 : M.wbits);
  return o == 128 ? 0 : -4;
}

// ASTC integer sequence encoding: T / Q words from the specification's decode equations (lowest word per tuple), built on the host
struct UIse { uint8_t trit[243]; uint8_t quint[125]; };
static void u_host_ise(UIse &I) {
  bool ht[243] = { false }, hq[125] = { false };
  for (int T = 0; T < 256; T++) {
    int t[5], C;
    if (((T >> 2) & 7) == 7) { C = ((T >> 5) << 2) | (T & 3); t[4] = t[3] = 2; }
    else { C = T & 31; if (((T >> 5) & 3) == 3) { t[4] = 2; t[3] = (T >> 7) & 1; } else { t[4] = (T >> 7) & 1; t[3] = (T >> 5) & 3; } }
    if ((C & 3) == 3) { t[2] = 2; t[1] = (C >> 4) & 1; t[0] = (((C >> 3) & 1) << 1) | (((C >> 2) & 1) & ~((C >> 3) & 1)); }
    else if (((C >> 2) & 3) == 3) { t[2] = 2; t[1] = 2; t[0] = C & 3; }
    else { t[2] = (C >> 4) & 1; t[1] = (C >> 2) & 3; t[0] = (((C >> 1) & 1) << 1) | ((C & 1) & ~((C >> 1) & 1)); }
    const int k = t[0] + 3 * t[1] + 9 * t[2] + 27 * t[3] + 81 * t[4];
    if (!ht[k]) { ht[k] = true; I.trit[k] = (uint8_t)T; }
  }
  for (int Q = 0; Q < 128; Q++) {
    int q[3];
    if (((Q >> 1) & 3) == 3 && ((Q >> 5) & 3) == 0) { q[2] = ((Q & 1) << 2) | ((((Q >> 4) & 1) & ~(Q & 1)) << 1) | (((Q >> 3) & 1) & ~(Q & 1)); q[1] = q[0] = 4; }
    else {
      int C;
      if (((Q >> 1) & 3) == 3) { q[2] = 4; C = (((Q >> 3) & 3) << 3) | ((~(Q >> 5) & 3) << 1) | (Q & 1); } else { q[2] = (Q >> 5) & 3; C = Q & 31; }
      if ((C & 7) == 5) { q[1] = 4; q[0] = (C >> 3) & 3; } else { q[1] = (C >> 3) & 3; q[0] = C & 7; }
    }
    const int k = q[0] + 5 * q[1] + 25 * q[2];
    if (q[0] < 5 && q[1] < 5 && q[2] < 5 && !hq[k]) { hq[k] = true; I.quint[k] = (uint8_t)Q; }
  }
}
struct UConst { UTab tab; UIse ise; };

// UASTC block -> ASTC 4x4 block
__device__ inline int u_to_astc(const UBits &B, const UConst *K, UBits &A) {
  ULog L; uint32_t solid = 0;
  for (int i = 0; i < 8; i++) L.ep[i] = 0;
  for (int i = 0; i < 32; i++) L.w[i] = 0;
  const int rc = u_unpack(B, L, solid);
  A.lo = 0; A.hi = 0;
  if (rc) return rc;
  if (L.mode == 8) {                                                     // LDR void extent, all-ones extent, 4 x UNORM16
    A.lo = 0xFFFFFFFFFFFFFDFCull;
    for (int c = 0; c < 4; c++) { const unsigned long long v = (solid >> (8 * c)) & 255u; A.hi |= ((v << 8) | v) << (16 * c); }
    return 0;
  }
  const UMode M = u_mode(L.mode); const int slot = u_slot(M.range), maxw = (1 << M.wbits) - 1;
  int s0 = 0, s1 = 0;
  for (int c = 0; c < 3; c++) { s0 += K->tab.uq[slot][L.ep[2 * c]]; s1 += K->tab.uq[slot][L.ep[2 * c + 1]]; }
  if (s1 < s0) {                                                         // ASTC would apply blue contraction: swap the pair, mirror the weights
    for (int c = 0; c < M.comps; c++) { const uint8_t t = L.ep[2 * c]; L.ep[2 * c] = L.ep[2 * c + 1]; L.ep[2 * c + 1] = t; }
    for (int i = 0; i < 16 * M.planes; i++) L.w[i] = (uint8_t)(maxw - L.w[i]);
  }
  int o = 0;
  A.put(o, M.astc_bm, 11); A.put(o, 0, 2); A.put(o, M.comps == 3 ? 8u : 12u, 4);
  const int nv = 2 * M.comps, bits = u_rbits(M.range);
  if (u_rtrit(M.range)) {
    const int sh[5] = { 0, 2, 4, 5, 7 }, nb[5] = { 2, 2, 1, 2, 1 };
    for (int g = 0; g < nv; g += 5) {
      int k = 0, s = 1; for (int j = 0; j < 5; j++, s *= 3) k += (g + j < nv ? L.ep[g + j] >> bits : 0) * s;
      const int Tw = K->ise.trit[k];
      for (int j = 0; j < 5 && g + j < nv; j++) { A.put(o, L.ep[g + j] & ((1u << bits) - 1u), bits); A.put(o, (uint32_t)(Tw >> sh[j]), nb[j]); }
    }
  } else if (u_rquint(M.range)) {
    const int sh[3] = { 0, 3, 5 }, nb[3] = { 3, 2, 2 };
    for (int g = 0; g < nv; g += 3) {
      int k = 0, s = 1; for (int j = 0; j < 3; j++, s *= 5) k += (g + j < nv ? L.ep[g + j] >> bits : 0) * s;
      const int Qw = K->ise.quint[k];
      for (int j = 0; j < 3 && g + j < nv; j++) { A.put(o, L.ep[g + j] & ((1u << bits) - 1u), bits); A.put(o, (uint32_t)(Qw >> sh[j]), nb[j]); }
    }
  } else for (int i = 0; i < nv; i++) A.put(o, L.ep[i], bits);
  const int wtot = 16 * M.planes * M.wbits;
  if (M.planes == 2) { int oc = 128 - wtot - 2; A.put(oc, (uint32_t)L.ccs, 2); }
  for (int i = 0; i < 16 * M.planes; i++) for (int b = 0; b < M.wbits; b++) if ((L.w[i] >> b) & 1) { const int pos = 127 - (i * M.wbits + b); if (pos < 64) A.lo |= 1ull << pos; else A.hi |= 1ull << (pos - 64); }
  return 0;
}

// ---- UASTC -> BC7 (target 2): what the stock loader asks a UASTC source for on every desktop GPU (reference src/lib/KTX2Loader.js:601-609,
// chosen at :665-676 when ASTC is not supported).  A deterministic re-fit of the block's own endpoints and decoded texels, gated by PSNR
// against the RGBA32 decode, not by bit parity with the basis transcoder (whose tables are not in the reference):
//  * single-plane modes (0, 10, 12, 18) and solid blocks -> BC7 mode 6 (RGBA 7-bit endpoints + a p-bit each, 4-bit indices): each endpoint
//    takes the p-bit under which its four 8-bit values are represented best, every texel the index of the nearest interpolated colour;
//  * dual-plane modes (6, 11) -> BC7 mode 5 (RGB 7-bit endpoints / 2-bit indices + a separate 8-bit scalar channel with its own indices;
//    the rotation puts the block's second-plane channel there): UASTC's and BC7's 2-bit weights are the same {0, 21, 43, 64}.
// Bit layouts as in tex_decode.hip (k_tdec_bc7), LSB first.
struct UBc7 { unsigned long long lo, hi; int pos;
  __device__ __forceinline__ void put(unsigned long long v, int n) { if (pos < 64) { lo |= v << pos; if (pos + n > 64) hi |= v >> (64 - pos); } else hi |= v << (pos - 64); pos += n; } };
__device__ inline void u_to_bc7(const ULog &L, uint32_t solid, const uint32_t px[16], const UTab *T, UBits &out) {
  const int W2[4] = { 0, 21, 43, 64 };
  const int W4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };
  int lo[4] = { 0, 0, 0, 255 }, hi[4] = { 0, 0, 0, 255 }, dual = 0, ccs = 0;
  if (L.mode == 8) { for (int c = 0; c < 4; c++) lo[c] = hi[c] = (int)((solid >> (8 * c)) & 255u); }
  else {
    const UMode M = u_mode(L.mode); const int slot = u_slot(M.range);
    for (int c = 0; c < M.comps; c++) { lo[c] = T->uq[slot][L.ep[2 * c]]; hi[c] = T->uq[slot][L.ep[2 * c + 1]]; }
    dual = M.planes == 2; ccs = L.ccs;
  }
  UBc7 B; B.lo = 0; B.hi = 0; B.pos = 0;
  if (!dual) {
    int e7[2][4], pb[2] = { 0, 0 };
    for (int s = 0; s < 2; s++) {
      int best = 1 << 30;
      for (int p = 0; p < 2; p++) {
        int q[4], err = 0;
        for (int c = 0; c < 4; c++) { const int v = s ? hi[c] : lo[c]; int t = (v - p + 1) >> 1; t = t < 0 ? 0 : (t > 127 ? 127 : t); q[c] = t; const int d = ((t << 1) | p) - v; err += d * d; }
        if (err < best) { best = err; pb[s] = p; for (int c = 0; c < 4; c++) e7[s][c] = q[c]; }
      }
    }
    uint32_t idx[16];
    for (int i = 0; i < 16; i++) {
      uint32_t bw = 0; int be = 1 << 30;
      for (int k = 0; k < 16; k++) {
        int err = 0;
        for (int c = 0; c < 4; c++) { const int a = (e7[0][c] << 1) | pb[0], b = (e7[1][c] << 1) | pb[1], d = ((a * (64 - W4[k]) + b * W4[k] + 32) >> 6) - (int)((px[i] >> (8 * c)) & 255u); err += d * d; }
        if (err < be) { be = err; bw = (uint32_t)k; }
      }
      idx[i] = bw;
    }
    const int swap = idx[0] >= 8u ? 1 : 0;
    B.lo = 1ull << 6; B.pos = 7;
    for (int c = 0; c < 4; c++) { B.put((unsigned)e7[swap][c], 7); B.put((unsigned)e7[swap ^ 1][c], 7); }
    B.put((unsigned)pb[swap], 1); B.put((unsigned)pb[swap ^ 1], 1);
    for (int i = 0; i < 16; i++) B.put(swap ? 15u - idx[i] : idx[i], i == 0 ? 3 : 4);
  } else {
    const int rot = ccs == 3 ? 0 : ccs + 1;
    int src[3]; for (int k = 0; k < 3; k++) src[k] = (ccs < 3 && k == ccs) ? 3 : k;       // encoded colour channel k holds this actual channel
    int e7[2][3];
    for (int s = 0; s < 2; s++) for (int k = 0; k < 3; k++) {
      const int v = s ? hi[src[k]] : lo[src[k]]; int bt = 0, bd = 1 << 30;
      for (int t = (v >> 1) - 1; t <= (v >> 1) + 1; t++) { if (t < 0 || t > 127) continue; int d = ((t << 1) | (t >> 6)) - v; d = d < 0 ? -d : d; if (d < bd) { bd = d; bt = t; } }
      e7[s][k] = bt;
    }
    const int a0 = lo[ccs], a1 = hi[ccs];
    uint32_t ci[16], ai[16];
    for (int i = 0; i < 16; i++) {
      uint32_t bw = 0; int be = 1 << 30;
      for (int k = 0; k < 4; k++) {
        int err = 0;
        for (int c = 0; c < 3; c++) { const int a = (e7[0][c] << 1) | (e7[0][c] >> 6), b = (e7[1][c] << 1) | (e7[1][c] >> 6), d = ((a * (64 - W2[k]) + b * W2[k] + 32) >> 6) - (int)((px[i] >> (8 * src[c])) & 255u); err += d * d; }
        if (err < be) { be = err; bw = (uint32_t)k; }
      }
      ci[i] = bw; bw = 0; be = 1 << 30;
      for (int k = 0; k < 4; k++) { const int d = ((a0 * (64 - W2[k]) + a1 * W2[k] + 32) >> 6) - (int)((px[i] >> (8 * ccs)) & 255u); if (d * d < be) { be = d * d; bw = (uint32_t)k; } }
      ai[i] = bw;
    }
    const int cswap = ci[0] >= 2u ? 1 : 0, aswap = ai[0] >= 2u ? 1 : 0;
    B.lo = 1ull << 5; B.pos = 6;
    B.put((unsigned)rot, 2);
    for (int c = 0; c < 3; c++) { B.put((unsigned)e7[cswap][c], 7); B.put((unsigned)e7[cswap ^ 1][c], 7); }
    B.put((unsigned)(aswap ? a1 : a0), 8); B.put((unsigned)(aswap ? a0 : a1), 8);
    for (int i = 0; i < 16; i++) B.put(cswap ? 3u - ci[i] : ci[i], i == 0 ? 1 : 2);
    for (int i = 0; i < 16; i++) B.put(aswap ? 3u - ai[i] : ai[i], i == 0 ? 1 : 2);
  }
  out.lo = B.lo; out.hi = B.hi;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
struct UastcJob {                       // one segment (= one .ktx2): layers of W x H RGBA8, top row first
  const uint8_t *layer[64]; uint8_t *out[64];      // encode: out[l] = UASTC blocks of layer l; decode: layer[l] = UASTC blocks, out[l] = target
  uint32_t W, H, L, bx, by; int32_t yflip, status, any_alpha;
};

// grid (blocks of 256 texel-blocks, layer, segment)
// four waves per SIMD (128 VGPRs): measured 75 / 63 / 58 / 64 / 67 / 79 ms per 120 layers of 2048^2 with the compiler's own choice (204 VGPRs,
// two waves) / 3 / 4 / 5 / 6 / 8 waves - the spills of the tighter budgets cost more than the extra waves hide
__global__ void __launch_bounds__(UVOL_BLOCK) UVOL_WAVES_PER_EU(4) k_uastc_encode(UastcJob *jobs, const UConst *K) {
  UastcJob &J = jobs[blockIdx.z];
  __shared__ UTab T;
  for (uint32_t i = threadIdx.x; i < sizeof(UTab) / 4; i += UVOL_BLOCK) reinterpret_cast<uint32_t *>(&T)[i] = reinterpret_cast<const uint32_t *>(&K->tab)[i];
  __syncthreads();
  const uint32_t l = blockIdx.y, b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (l >= J.L || b >= J.bx * J.by) return;
  const uint32_t X = b % J.bx, Y = b / J.bx;
  const uint8_t *img = J.layer[l];
  uint32_t px[16]; bool alpha = false;
  for (int y = 0; y < 4; y++) {
    uint32_t py = Y * 4 + (uint32_t)y; if (py >= J.H) py = J.H - 1;
    const uint32_t sr = J.yflip ? J.H - 1 - py : py;
    const uint8_t *row = img + 4 * ((size_t)sr * J.W);
    if (X * 4 + 3 < J.W && (J.W & 3) == 0) { const uint4 v = *reinterpret_cast<const uint4 *>(row + 16 * (size_t)X); px[4 * y] = v.x; px[4 * y + 1] = v.y; px[4 * y + 2] = v.z; px[4 * y + 3] = v.w; }
    else for (int x = 0; x < 4; x++) { uint32_t pxx = X * 4 + (uint32_t)x; if (pxx >= J.W) pxx = J.W - 1; const uint8_t *p = row + 4 * (size_t)pxx; px[4 * y + x] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
  }
  for (int i = 0; i < 16; i++) alpha |= (px[i] >> 24) != 255u;
  if (alpha) J.any_alpha = 1;                                            // the container's DFD channel id says RGBA when any block carries alpha
  UBits B; u_encode_block(px, &T, B);
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(J.out[l] + 16 * (size_t)b);
  dst[0] = B.lo; dst[1] = B.hi;
}
// target 0: RGBA8 (W x H x 4 per layer, stored row order), 1: ASTC 4x4 blocks, 2: BC7 blocks
// EAC alpha modifier tables (ETC2 RGBA8's alpha half; the same values as tex_decode.hip's EAC_MOD)
__device__ const int8_t UASTC_EAC_MOD[16][8] = {
  { -3, -6, -9, -15, 2, 5, 8, 14 }, { -3, -7, -10, -13, 2, 6, 9, 12 }, { -2, -5, -8, -13, 1, 4, 7, 12 }, { -2, -4, -6, -13, 1, 3, 5, 12 },
  { -3, -6, -8, -12, 2, 5, 7, 11 }, { -3, -7, -9, -11, 2, 6, 8, 10 }, { -4, -7, -8, -11, 3, 6, 7, 10 }, { -3, -5, -8, -11, 2, 4, 7, 10 },
  { -2, -6, -8, -10, 1, 5, 7, 9 }, { -2, -5, -8, -10, 1, 4, 7, 9 }, { -2, -4, -8, -10, 1, 3, 7, 9 }, { -2, -5, -7, -10, 1, 4, 6, 9 },
  { -3, -4, -7, -10, 2, 3, 6, 9 }, { -1, -2, -3, -10, 0, 1, 2, 9 }, { -4, -6, -8, -9, 3, 5, 7, 8 }, { -3, -5, -7, -9, 2, 4, 6, 8 } };
__global__ void __launch_bounds__(UVOL_BLOCK) k_uastc_decode(UastcJob *jobs, const UConst *K, int target) {
  UastcJob &J = jobs[blockIdx.z];
  const uint32_t l = blockIdx.y, b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (l >= J.L || b >= J.bx * J.by) return;
  const unsigned long long *src = reinterpret_cast<const unsigned long long *>(J.layer[l] + 16 * (size_t)b);
  UBits B; B.lo = src[0]; B.hi = src[1];
  if (target == 1) {
    UBits A; if (u_to_astc(B, K, A)) J.status = -10;
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(J.out[l] + 16 * (size_t)b);
    dst[0] = A.lo; dst[1] = A.hi; return;
  }
  ULog L; uint32_t solid = 0, px[16];
  for (int i = 0; i < 8; i++) L.ep[i] = 0;
  for (int i = 0; i < 32; i++) L.w[i] = 0;
  if (u_unpack(B, L, solid)) { J.status = -10; return; }
  if (L.mode == 8) for (int i = 0; i < 16; i++) px[i] = solid; else u_decode_log(L, &K->tab, px);
  if (target == 2) {
    UBits A; u_to_bc7(L, solid, px, &K->tab, A);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(J.out[l] + 16 * (size_t)b);
    dst[0] = A.lo; dst[1] = A.hi; return;
  }
  if (target == 5 || target == 6) {
    // ETC1 (8 bytes) / ETC2 RGBA (16 bytes: EAC alpha block, then the colour block) from the decoded texels (round 5; the stock loader's etc2Supported /
    // etc1Supported rows for UASTC sources, src/lib/KTX2Loader.js:619-636: what a UASTC file is asked for on ETC2 hardware without ASTC).  A plain
    // ETC1 fit, not the basis transcoder's hint-driven one (its tables are not in the reference): both flips are tried; a half-block's base is its
    // mean colour, in differential mode (5 bits + a 3-bit delta clamped to [-4, 3], which also keeps the block out of ETC2's T / H / planar modes)
    // when both deltas fit, else in individual mode (4 bits each); per half-block the intensity table with the smallest error under per-texel
    // optimal modifiers; the flip with the smaller total error wins (ties: flip 0).  Gated by PSNR against the RGBA32 decode.
    const int MAG[8][2] = { { 2, 8 }, { 5, 17 }, { 9, 29 }, { 13, 42 }, { 18, 60 }, { 24, 80 }, { 33, 106 }, { 47, 183 } };
    uint8_t *out = J.out[l] + (target == 6 ? 16 : 8) * (size_t)b;
    if (target == 6) {
      int a0 = 0, a1 = 255;
      for (int i = 0; i < 16; i++) { const int a = (int)(px[i] >> 24); a0 = a > a0 ? a : a0; a1 = a < a1 ? a : a1; }
      int bt = 13, bm = 1, bb = a0;                       // constant alpha: table 13 has a zero modifier (index 4)
      if (a0 > a1) {
        const int mid = (a0 + a1 + 1) >> 1; uint32_t best = 0xffffffffu;
        for (int t = 0; t < 16 && best; t++) for (int m = 1; m < 16 && best; m++) for (int db = -2; db <= 2; db++) {
          const int base = mid + db; if (base < 0 || base > 255) continue;
          int lv[8]; for (int j = 0; j < 8; j++) { const int v = base + m * UASTC_EAC_MOD[t][j]; lv[j] = v < 0 ? 0 : (v > 255 ? 255 : v); }
          uint32_t err = 0;
          for (int i = 0; i < 16 && err < best; i++) { const int a = (int)(px[i] >> 24); int be = 1 << 30; for (int j = 0; j < 8; j++) { const int d = lv[j] - a; be = d * d < be ? d * d : be; } err += (uint32_t)be; }
          if (err < best) { best = err; bt = t; bm = m; bb = base; }
        }
      }
      unsigned long long bits = 0;
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
        const int a = (int)(px[4 * y + x] >> 24); int be = 1 << 30; uint32_t bj = 0;
        for (int j = 0; j < 8; j++) { int v = bb + bm * UASTC_EAC_MOD[bt][j]; v = v < 0 ? 0 : (v > 255 ? 255 : v); const int d = v - a; if (d * d < be) { be = d * d; bj = (uint32_t)j; } }
        bits |= (unsigned long long)bj << (45 - 3 * (4 * x + y));
      }
      out[0] = (uint8_t)bb; out[1] = (uint8_t)((bm << 4) | bt);
      for (int k = 0; k < 6; k++) out[2 + k] = (uint8_t)(bits >> (40 - 8 * k));
      out += 8;
    }
    uint32_t best_err = 0xffffffffu; uint8_t best_blk[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int flip = 0; flip < 2; flip++) {
      int sum[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } };
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { const int h = flip ? (y >= 2) : (x >= 2); for (int c = 0; c < 3; c++) sum[h][c] += (int)((px[4 * y + x] >> (8 * c)) & 255u); }
      int q5[2][3], q4[2][3]; bool diff = true;
      for (int h = 0; h < 2; h++) for (int c = 0; c < 3; c++) { const int m8 = (sum[h][c] + 4) >> 3; q5[h][c] = (m8 * 31 + 127) / 255; q4[h][c] = (m8 * 15 + 127) / 255; }
      for (int c = 0; c < 3; c++) { const int d = q5[1][c] - q5[0][c]; if (d < -4 || d > 3) diff = false; }
      int base[2][3];
      for (int h = 0; h < 2; h++) for (int c = 0; c < 3; c++) base[h][c] = diff ? ((q5[h][c] << 3) | (q5[h][c] >> 2)) : q4[h][c] * 17;
      uint32_t err = 0; int tab[2] = { 0, 0 }; uint32_t msb = 0, lsb = 0;
      for (int h = 0; h < 2; h++) {
        uint32_t bh = 0xffffffffu; uint32_t bm_ = 0, bl_ = 0;
        for (int t = 0; t < 8; t++) {
          uint32_t e = 0, m_ = 0, l_ = 0;
          for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
            if ((flip ? (y >= 2) : (x >= 2)) != (h != 0)) continue;
            const uint32_t p = px[4 * y + x]; int be = 1 << 30; uint32_t bi = 0;
            for (uint32_t idx = 0; idx < 4; idx++) {
              const int mod = (idx & 2u) ? -MAG[t][idx & 1u] : MAG[t][idx & 1u]; int d = 0;
              for (int c = 0; c < 3; c++) { int v = base[h][c] + mod; v = v < 0 ? 0 : (v > 255 ? 255 : v); const int q = v - (int)((p >> (8 * c)) & 255u); d += q * q; }
              if (d < be) { be = d; bi = idx; }
            }
            e += (uint32_t)be; const int i = 4 * x + y; m_ |= (bi >> 1) << i; l_ |= (bi & 1u) << i;
          }
          if (e < bh) { bh = e; tab[h] = t; bm_ = m_; bl_ = l_; }
        }
        err += bh; msb |= bm_; lsb |= bl_;
      }
      if (err < best_err) {
        best_err = err;
        for (int c = 0; c < 3; c++) best_blk[c] = diff ? (uint8_t)((q5[0][c] << 3) | ((q5[1][c] - q5[0][c]) & 7)) : (uint8_t)((q4[0][c] << 4) | q4[1][c]);
        best_blk[3] = (uint8_t)((tab[0] << 5) | (tab[1] << 2) | (diff ? 2 : 0) | flip);
        best_blk[4] = (uint8_t)(msb >> 8); best_blk[5] = (uint8_t)msb; best_blk[6] = (uint8_t)(lsb >> 8); best_blk[7] = (uint8_t)lsb;
      }
    }
    for (int k = 0; k < 8; k++) out[k] = best_blk[k];
    return;
  }
  if (target == 3 || target == 4) {
    // BC1 (8 bytes) / BC3 (16 bytes) from the decoded texels (round 5; the stock loader's dxtSupported row for UASTC sources,
    // src/lib/KTX2Loader.js:610-618).  Colour: range fit - the corners of the texels' bounding box, the R and B ends swapped where the channel
    // runs against G (sign of the covariance), pulled in by 1/16 of the range, rounded to RGB565; every texel takes the nearest of the four
    // palette colours; an opaque BC1 block needs colour0 > colour1 (swap + remap, equal endpoints: index 0).  Alpha (BC3): a BC4 block with
    // alpha0 = the largest, alpha1 = the smallest alpha of the block, eight-value mode.  Re-fits gated by PSNR against the RGBA32 decode.
    uint8_t *out = J.out[l] + (target == 4 ? 16 : 8) * (size_t)b;
    if (target == 4) {
      int a0 = 0, a1 = 255;
      for (int i = 0; i < 16; i++) { const int a = (int)(px[i] >> 24); a0 = a > a0 ? a : a0; a1 = a < a1 ? a : a1; }
      unsigned long long bits = 0;
      if (a0 > a1) for (int i = 0; i < 16; i++) {
        const int a = (int)(px[i] >> 24); int be = 1 << 30; uint32_t bj = 0;
        for (int j = 0; j < 8; j++) { const int v = j == 0 ? a0 : (j == 1 ? a1 : ((8 - j) * a0 + (j - 1) * a1) / 7); const int d = v - a; if (d * d < be) { be = d * d; bj = (uint32_t)j; } }
        bits |= (unsigned long long)bj << (3 * i);
      }
      out[0] = (uint8_t)a0; out[1] = (uint8_t)a1;
      for (int k = 0; k < 6; k++) out[2 + k] = (uint8_t)(bits >> (8 * k));
      out += 8;
    }
    int mn[3] = { 255, 255, 255 }, mx[3] = { 0, 0, 0 }, sum[3] = { 0, 0, 0 };
    for (int i = 0; i < 16; i++) for (int c = 0; c < 3; c++) { const int v = (int)((px[i] >> (8 * c)) & 255u); mn[c] = v < mn[c] ? v : mn[c]; mx[c] = v > mx[c] ? v : mx[c]; sum[c] += v; }
    int cov_rg = 0, cov_bg = 0;
    for (int i = 0; i < 16; i++) {
      const int r = 16 * (int)(px[i] & 255u) - sum[0], g = 16 * (int)((px[i] >> 8) & 255u) - sum[1], bl = 16 * (int)((px[i] >> 16) & 255u) - sum[2];
      cov_rg += (r >> 4) * (g >> 4); cov_bg += (bl >> 4) * (g >> 4);
    }
    int hi[3] = { mx[0], mx[1], mx[2] }, lo[3] = { mn[0], mn[1], mn[2] };
    if (cov_rg < 0) { hi[0] = mn[0]; lo[0] = mx[0]; }
    if (cov_bg < 0) { hi[2] = mn[2]; lo[2] = mx[2]; }
    for (int c = 0; c < 3; c++) { const int ins = (hi[c] - lo[c]) / 16; hi[c] -= ins; lo[c] += ins; }
    const int q0[3] = { (hi[0] * 31 + 127) / 255, (hi[1] * 63 + 127) / 255, (hi[2] * 31 + 127) / 255 };
    const int q1[3] = { (lo[0] * 31 + 127) / 255, (lo[1] * 63 + 127) / 255, (lo[2] * 31 + 127) / 255 };
    uint32_t c0 = (uint32_t)((q0[0] << 11) | (q0[1] << 5) | q0[2]), c1 = (uint32_t)((q1[0] << 11) | (q1[1] << 5) | q1[2]);
    int pal[4][3];
    { const int e0[3] = { (q0[0] << 3) | (q0[0] >> 2), (q0[1] << 2) | (q0[1] >> 4), (q0[2] << 3) | (q0[2] >> 2) };
      const int e1[3] = { (q1[0] << 3) | (q1[0] >> 2), (q1[1] << 2) | (q1[1] >> 4), (q1[2] << 3) | (q1[2] >> 2) };
      for (int c = 0; c < 3; c++) { pal[0][c] = e0[c]; pal[1][c] = e1[c]; pal[2][c] = (2 * e0[c] + e1[c]) / 3; pal[3][c] = (e0[c] + 2 * e1[c]) / 3; } }
    uint32_t idx = 0;
    if (c0 != c1) {
      const uint32_t flip = c0 < c1 ? 1u : 0u;             // four-colour mode wants colour0 > colour1: swap, 0 <-> 1 and 2 <-> 3
      for (int i = 0; i < 16; i++) {
        int be = 1 << 30; uint32_t bj = 0;
        for (int j = 0; j < 4; j++) { int d = 0; for (int c = 0; c < 3; c++) { const int t = pal[j][c] - (int)((px[i] >> (8 * c)) & 255u); d += t * t; } if (d < be) { be = d; bj = (uint32_t)j; } }
        idx |= (bj ^ flip) << (2 * i);
      }
      if (flip) { const uint32_t t = c0; c0 = c1; c1 = t; }
    }
    out[0] = (uint8_t)c0; out[1] = (uint8_t)(c0 >> 8); out[2] = (uint8_t)c1; out[3] = (uint8_t)(c1 >> 8);
    out[4] = (uint8_t)idx; out[5] = (uint8_t)(idx >> 8); out[6] = (uint8_t)(idx >> 16); out[7] = (uint8_t)(idx >> 24);
    return;
  }
  const uint32_t X = b % J.bx, Y = b / J.bx;
  for (int y = 0; y < 4 && 4 * Y + (uint32_t)y < J.H; y++) {
    uint8_t *row = J.out[l] + 4 * ((size_t)(4 * Y + (uint32_t)y) * J.W + 4 * (size_t)X);
    if (X * 4 + 3 < J.W && (J.W & 3) == 0) *reinterpret_cast<uint4 *>(row) = make_uint4(px[4 * y], px[4 * y + 1], px[4 * y + 2], px[4 * y + 3]);
    else for (int x = 0; x < 4 && 4 * X + (uint32_t)x < J.W; x++) { const uint32_t v = px[4 * y + x]; row[4 * x] = (uint8_t)v; row[4 * x + 1] = (uint8_t)(v >> 8); row[4 * x + 2] = (uint8_t)(v >> 16); row[4 * x + 3] = (uint8_t)(v >> 24); }
  }
}

// ================================================================================================
// host side
// ================================================================================================
struct UastcState { uvol_devbuf consts, jobs, layers, blocks, outs; std::vector<UastcJob> hjobs; uint8_t *pinned = nullptr; size_t pinned_cap = 0; bool ready = false; };

int uastc_create(uvol_ctx *ctx) { ctx->uastc = new UastcState(); return UVOL_OK; }
void uastc_destroy(uvol_ctx *ctx) {
  UastcState *U = ctx->uastc; if (!U) return;
  for (uvol_devbuf *b : { &U->consts, &U->jobs, &U->layers, &U->blocks, &U->outs }) if (b->p) (void)hipFree(b->p);
  if (U->pinned) (void)hipHostFree(U->pinned);
  delete U; ctx->uastc = nullptr;
}
static int uastc_consts(uvol_ctx *ctx) {
  UastcState *U = ctx->uastc;
  if (U->ready) return UVOL_OK;
  static UConst K; static bool built = false;
  if (!built) { u_host_tables(K.tab); memset(&K.ise, 0, sizeof K.ise); u_host_ise(K.ise); built = true; }
  if (int rc = uvol_ensure(ctx, U->consts, sizeof(UConst))) return rc;
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(U->consts.p, &K, sizeof(UConst), hipMemcpyHostToDevice, ctx->stream));
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  U->ready = true;
  return UVOL_OK;
}
static int uastc_pinned(uvol_ctx *ctx, size_t bytes) {
  UastcState *U = ctx->uastc;
  if (bytes <= U->pinned_cap) return UVOL_OK;
  if (U->pinned) (void)hipHostFree(U->pinned);
  U->pinned = nullptr; U->pinned_cap = 0;
  UVOL_HIP_CHECK(ctx, hipHostMalloc((void **)&U->pinned, bytes + bytes / 8 + 4096, hipHostMallocDefault));
  U->pinned_cap = bytes + bytes / 8 + 4096;
  return UVOL_OK;
}
namespace {
inline void uput32(uint8_t *&p, uint32_t v) { memcpy(p, &v, 4); p += 4; }
inline void uput64(uint8_t *&p, uint64_t v) { memcpy(p, &v, 8); p += 8; }
inline void uput16(uint8_t *&p, uint16_t v) { memcpy(p, &v, 2); p += 2; }
// KTX2 header + DFD (colour model 166 = UASTC, BT.709, sRGB, 16-byte 4x4 texel blocks) + key/value data; returns the level offset
size_t uastc_header(uint8_t *out, uint32_t W, uint32_t H, uint32_t L, bool alpha, uint64_t lvl_len) {
  static const uint8_t ident[12] = { 0xAB, 'K', 'T', 'X', ' ', '2', '0', 0xBB, '\r', '\n', 0x1A, '\n' };
  static const char writer[] = "uvol-mi355x uastc 0.1";
  uint8_t kvd[128]; uint8_t *kp = kvd;
  uput32(kp, 12 + 12); memcpy(kp, "KTXanimData", 12); kp += 12; uput32(kp, 1); uput32(kp, 15); uput32(kp, 0);
  uput32(kp, 10 + (uint32_t)sizeof(writer)); memcpy(kp, "KTXwriter", 10); kp += 10; memcpy(kp, writer, sizeof(writer)); kp += sizeof(writer);
  while ((kp - kvd) & 3) *kp++ = 0;
  const uint32_t dfd_off = 80 + 24, dfd_len = 44, kvd_off = dfd_off + dfd_len, kvd_len = (uint32_t)(kp - kvd);
  const uint64_t lvl_off = ((uint64_t)kvd_off + kvd_len + 15) & ~15ull;
  if (!out) return (size_t)lvl_off;
  uint8_t *p = out;
  memcpy(p, ident, 12); p += 12;
  uput32(p, 0); uput32(p, 1); uput32(p, W); uput32(p, H); uput32(p, 0); uput32(p, L); uput32(p, 1); uput32(p, 1); uput32(p, 0);
  uput32(p, dfd_off); uput32(p, dfd_len); uput32(p, kvd_off); uput32(p, kvd_len); uput64(p, 0); uput64(p, 0);
  uput64(p, lvl_off); uput64(p, lvl_len); uput64(p, lvl_len);
  uput32(p, 44); uput32(p, 0); uput16(p, 2); uput16(p, 40);
  *p++ = 166; *p++ = 1; *p++ = 2; *p++ = 0; *p++ = 3; *p++ = 3; *p++ = 0; *p++ = 0;
  *p++ = 16; for (int i = 0; i < 7; i++) *p++ = 0;
  uput16(p, 0); *p++ = 127; *p++ = alpha ? 3 : 0; *p++ = 0; *p++ = 0; *p++ = 0; *p++ = 0; uput32(p, 0); uput32(p, 0xFFFFFFFFu);
  memcpy(p, kvd, kvd_len); p += kvd_len;
  while ((uint64_t)(p - out) < lvl_off) *p++ = 0;
  return (size_t)lvl_off;
}
}  // namespace

// container fields of a UASTC .ktx2 this codec reads (vkFormat 0, DFD model 166, no supercompression, 1 level, 1 face)
int uastc_ktx2_probe(const uint8_t *b, size_t n, uint32_t *W, uint32_t *H, uint32_t *L, uint64_t *lvl_off) {
  static const uint8_t ident[12] = { 0xAB, 'K', 'T', 'X', ' ', '2', '0', 0xBB, '\r', '\n', 0x1A, '\n' };
  if (!b || n < 104 + 44 || n > 0xffffffffull * 16 || memcmp(b, ident, 12)) return -1;
  uint32_t u[9]; memcpy(u, b + 12, 36);
  uint32_t dfd_off, dfd_len; memcpy(&dfd_off, b + 48, 4); memcpy(&dfd_len, b + 52, 4);
  const bool model_uastc = dfd_len >= 44 && dfd_off <= n && dfd_len <= n - dfd_off && b[dfd_off + 12] == 166;
  // stock `basisu -uastc -ktx2` writes Zstandard-supercompressed level data (scheme 2) by default: recognised, not decoded (no Zstd here)
  if (model_uastc && u[0] == 0 && u[8] != 0) return UASTC_PROBE_SUPERCOMPRESSED;
  if (u[0] != 0 || u[8] != 0 || u[7] != 1 || u[6] != 1 || !u[2] || !u[3] || u[2] > 16384 || u[3] > 16384) return -2;
  if (!model_uastc) return -3;
  uint64_t lo, ll; memcpy(&lo, b + 80, 8); memcpy(&ll, b + 88, 8);
  const uint32_t layers = u[5] ? u[5] : 1;
  if (layers > 64) return -4;
  const uint64_t need = (uint64_t)layers * ((u[2] + 3) / 4) * ((u[3] + 3) / 4) * 16;
  if (lo > n || ll > n - lo || ll != need) return -4;
  *W = u[2]; *H = u[3]; *L = layers; *lvl_off = lo;
  return 0;
}

// Zstandard-supercompressed UASTC (supercompressionScheme 2: what stock `basisu -uastc -ktx2` writes by default; KTX 2.0: every level's data
// is one Zstandard frame, levelIndex.byteLength = compressed, .uncompressedByteLength = original size, no supercompression global data).
// Round 5: read when the system's libzstd is installed (dlopen, like libdeflate for PNGs - no Zstandard code here): the level is inflated on
// the host into an equivalent scheme-0 file (same header / DFD / key-values, level data at a 16-byte aligned offset) that the decoders then
// take as any other.  0: `out` holds that file; 1: no libzstd on this machine (the caller refuses the file as before); < 0: not such a file /
// corrupt frame / sizes that do not match the header.
// header of such a file: sizes and the level's place, checked against each other (no inflation)
static int uastc_zstd_header(const uint8_t *b, size_t n, uint32_t &w, uint32_t &h, uint32_t &layers, uint64_t &lo, uint64_t &ll, uint64_t &need) {
  if (!b || n < 104 + 44) return -1;
  uint32_t u[9]; memcpy(u, b + 12, 36);
  if (u[0] != 0 || u[8] != 2 || u[7] != 1 || u[6] != 1 || !u[2] || !u[3] || u[2] > 16384 || u[3] > 16384) return -2;      // scheme 2 = Zstandard; 1 level, 1 face
  uint64_t lu; memcpy(&lo, b + 80, 8); memcpy(&ll, b + 88, 8); memcpy(&lu, b + 96, 8);
  layers = u[5] ? u[5] : 1; w = u[2]; h = u[3];
  if (layers > 64) return -4;
  need = (uint64_t)layers * ((u[2] + 3) / 4) * ((u[3] + 3) / 4) * 16;
  if (lo < 104 || lo > n || ll > n - lo || lu != need) return -4;
  return 0;
}
// width / height / layers of a Zstandard-supercompressed UASTC file from its header alone (uvol_ktx2_info: nothing is inflated to report a size)
int uastc_zstd_info(const uint8_t *b, size_t n, uint32_t *W, uint32_t *H, uint32_t *L) {
  uint32_t w, h, l; uint64_t lo, ll, need;
  const int r = uastc_zstd_header(b, n, w, h, l, lo, ll, need); if (r) return r;
  *W = w; *H = h; *L = l; return 0;
}
int uastc_unzstd(const uint8_t *b, size_t n, std::vector<uint8_t> &out) {
  typedef size_t (*dec_fn)(void *, size_t, const void *, size_t); typedef unsigned (*err_fn)(size_t); typedef unsigned long long (*fcs_fn)(const void *, size_t);
  static dec_fn dec = nullptr; static err_fn is_err = nullptr; static fcs_fn fcs = nullptr;
  static const bool have = [] {
    void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL); if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL); if (!h) return false;
    dec = (dec_fn)dlsym(h, "ZSTD_decompress"); is_err = (err_fn)dlsym(h, "ZSTD_isError"); fcs = (fcs_fn)dlsym(h, "ZSTD_getFrameContentSize"); return dec && is_err && fcs; }();
  uint32_t w, h, layers; uint64_t lo, ll, need;
  const int rh = uastc_zstd_header(b, n, w, h, layers, lo, ll, need); if (rh) return rh;
  if (!have) return 1;
  // `need` comes from header fields a crafted 150-byte file controls (up to 64 layers x 16384^2 texels = 17 GB): nothing is allocated before
  // the FRAME says the same size (one-shot ZSTD_compress, which basisu uses, records it), and never more than a Zstandard frame of `ll` bytes can hold
  const unsigned long long fsz = fcs(b + lo, (size_t)ll);
  if (fsz != need || need > (uint64_t)ll * 65536ull + 65536ull) return -5;
  const size_t head = (size_t)lo, off = (head + 15) & ~(size_t)15;
  try { out.assign(off + (size_t)need, 0); } catch (...) { out.clear(); return -6; }
  memcpy(out.data(), b, head);
  const size_t got = dec(out.data() + off, (size_t)need, b + lo, (size_t)ll);
  if (is_err(got) || got != need) return -5;
  const uint32_t scheme0 = 0; const uint64_t off64 = off;
  memcpy(out.data() + 12 + 32, &scheme0, 4); memcpy(out.data() + 80, &off64, 8); memcpy(out.data() + 88, &need, 8); memcpy(out.data() + 96, &need, 8);
  return 0;
}

// n_seg segments of n_layers layers -> UASTC .ktx2 files (what `basisu -uastc -ktx2 -tex_type video` writes, without Zstandard)
int tex_uastc_encode_segments(uvol_ctx *ctx, const uint8_t *const *rgba, int n_seg, int n_layers, uint32_t W, uint32_t H,
                              bool on_device, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status) {
  if (on_device) { const int ro = png_order_before(ctx, ctx->stream, rgba, (size_t)n_seg * n_layers); if (ro != UVOL_OK) return ro; }      // layers un-filtered on the ingest stream
  UastcState *U = ctx->uastc;
  if (n_seg <= 0) return UVOL_OK;
  if (n_layers > 64 || W > 16384 || H > 16384 || n_seg > 65535) { ctx->set_error("texture segment: unsupported size"); return UVOL_E_UNSUPPORTED; }
  int rc; if ((rc = uastc_consts(ctx))) return rc;
  const uint32_t bx = (W + 3) / 4, by = (H + 3) / 4; const size_t nb = (size_t)bx * by, lbytes = (size_t)W * H * 4, seg_bytes = nb * 16 * (size_t)n_layers;
  if ((rc = uvol_ensure(ctx, U->jobs, sizeof(UastcJob) * (size_t)n_seg))) return rc;
  if ((rc = uvol_ensure(ctx, U->blocks, seg_bytes * (size_t)n_seg))) return rc;
  if (!on_device && (rc = uvol_ensure(ctx, U->layers, lbytes * (size_t)n_layers * (size_t)n_seg))) return rc;
  U->hjobs.assign((size_t)n_seg, UastcJob{});
  std::vector<UvolUpItem> ups;
  for (int s = 0; s < n_seg; s++) {
    UastcJob &J = U->hjobs[s]; J.W = W; J.H = H; J.L = (uint32_t)n_layers; J.bx = bx; J.by = by; J.yflip = ctx->prm.y_flip ? 1 : 0;
    for (int l = 0; l < n_layers; l++) {
      const uint8_t *src = rgba[(size_t)s * n_layers + l];
      if (on_device) J.layer[l] = src;
      else { uint8_t *d = (uint8_t *)U->layers.p + lbytes * ((size_t)s * n_layers + l); ups.push_back(UvolUpItem{ lbytes * ((size_t)s * n_layers + l), src, lbytes }); J.layer[l] = d; }
      J.out[l] = (uint8_t *)U->blocks.p + seg_bytes * (size_t)s + nb * 16 * (size_t)l;
    }
  }
  if (!on_device) { const int rcu = uvol_upload_staged(ctx, (uint8_t *)U->layers.p, ups); if (rcu != UVOL_OK) return rcu; }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(U->jobs.p, U->hjobs.data(), sizeof(UastcJob) * (size_t)n_seg, hipMemcpyHostToDevice, ctx->stream));
  { uvol_ctx::Scope sc(ctx, "tex.uastc_encode", (uint64_t)(lbytes + nb * 16) * n_layers * (uint64_t)n_seg);
    hipLaunchKernelGGL(k_uastc_encode, dim3(uvol_blocks(nb), (unsigned)n_layers, (unsigned)n_seg), dim3(UVOL_BLOCK), 0, ctx->stream, (UastcJob *)U->jobs.p, (const UConst *)U->consts.p); }
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  if ((rc = uastc_pinned(ctx, seg_bytes * (size_t)n_seg))) return rc;
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(U->hjobs.data(), U->jobs.p, sizeof(UastcJob) * (size_t)n_seg, hipMemcpyDeviceToHost, ctx->stream));
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(U->pinned, U->blocks.p, seg_bytes * (size_t)n_seg, hipMemcpyDeviceToHost, ctx->stream));
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->resolve_profile();
  int worst = UVOL_OK;
  const size_t lvl_off = uastc_header(nullptr, W, H, (uint32_t)n_layers, false, seg_bytes);
  for (int s = 0; s < n_seg; s++) {
    out_lens[s] = lvl_off + seg_bytes;
    if (status) status[s] = UVOL_OK;
    if (out_lens[s] > caps[s]) { ctx->set_error("texture segment %d: output buffer too small (%zu > %zu)", s, out_lens[s], caps[s]); worst = UVOL_E_NOSPACE; if (status) status[s] = UVOL_E_NOSPACE; continue; }
    (void)uastc_header(outs[s], W, H, (uint32_t)n_layers, U->hjobs[s].any_alpha != 0, seg_bytes);
    memcpy(outs[s] + lvl_off, U->pinned + seg_bytes * (size_t)s, seg_bytes);
  }
  return status ? UVOL_OK : worst;
}

// UASTC .ktx2 files -> target 0: RGBA8 layers, target 3: ASTC 4x4 blocks (layer buffers of the caller, host or device)
int tex_uastc_decode_segments(uvol_ctx *ctx, const uint8_t *const *files, const size_t *lens, int n, uint8_t *const *outp, size_t layer_cap, bool outputs_on_device, int target, int *status) {
  UastcState *U = ctx->uastc;
  if (n <= 0) return UVOL_OK;
  if (target < 0 || target > 6) { ctx->set_error("unknown transcode target %d", target); return UVOL_E_UNSUPPORTED; }
  int rc; if ((rc = uastc_consts(ctx))) return rc;
  uint32_t W = 0, H = 0, L = 0; uint64_t lo = 0;
  if (uastc_ktx2_probe(files[0], lens[0], &W, &H, &L, &lo)) { ctx->set_error("segment 0: not a UASTC .ktx2 this decoder handles"); return UVOL_E_INVALID; }
  const uint32_t bx = (W + 3) / 4, by = (H + 3) / 4; const size_t nb = (size_t)bx * by, seg_bytes = nb * 16 * (size_t)L;
  const size_t layer_bytes = target == 0 ? (size_t)W * H * 4 : ((target == 5 || target == 1) ? nb * 8 : nb * 16);
  if (layer_cap < layer_bytes) { ctx->set_error("layer buffers too small (%zu < %zu)", layer_cap, layer_bytes); return UVOL_E_NOSPACE; }
  if ((rc = uvol_ensure(ctx, U->jobs, sizeof(UastcJob) * (size_t)n))) return rc;
  if ((rc = uvol_ensure(ctx, U->blocks, seg_bytes * (size_t)n))) return rc;
  if (!outputs_on_device && (rc = uvol_ensure(ctx, U->outs, layer_bytes * (size_t)L * (size_t)n))) return rc;
  U->hjobs.assign((size_t)n, UastcJob{});
  for (int s = 0; s < n; s++) {
    uint32_t w2, h2, l2; uint64_t lo2;
    if (uastc_ktx2_probe(files[s], lens[s], &w2, &h2, &l2, &lo2) || w2 != W || h2 != H || l2 != L) { ctx->set_error("segment %d: not a UASTC .ktx2 of the batch's size", s); return UVOL_E_INVALID; }
    UVOL_HIP_CHECK(ctx, hipMemcpyAsync((uint8_t *)U->blocks.p + seg_bytes * (size_t)s, files[s] + lo2, seg_bytes, hipMemcpyHostToDevice, ctx->stream));
    UastcJob &J = U->hjobs[s]; J.W = W; J.H = H; J.L = L; J.bx = bx; J.by = by;
    for (uint32_t l = 0; l < L; l++) {
      J.layer[l] = (const uint8_t *)U->blocks.p + seg_bytes * (size_t)s + nb * 16 * (size_t)l;
      J.out[l] = outputs_on_device ? outp[(size_t)s * L + l] : (uint8_t *)U->outs.p + layer_bytes * ((size_t)s * L + l);
    }
  }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(U->jobs.p, U->hjobs.data(), sizeof(UastcJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  { uvol_ctx::Scope sc(ctx, target == 0 ? "texdec.uastc_rgba" : (target == 3 ? "texdec.uastc_astc" : (target == 2 ? "texdec.uastc_bc7" : ((target == 5 || target == 6) ? "texdec.uastc_bc13" : "texdec.uastc_etc"))), (uint64_t)(nb * 16 + layer_bytes) * L * (uint64_t)n);
    hipLaunchKernelGGL(k_uastc_decode, dim3(uvol_blocks(nb), L, (unsigned)n), dim3(UVOL_BLOCK), 0, ctx->stream, (UastcJob *)U->jobs.p, (const UConst *)U->consts.p, target == 0 ? 0 : (target == 3 ? 1 : (target == 2 ? 2 : (target == 5 ? 3 : (target == 6 ? 4 : (target == 1 ? 5 : 6)))))); }      // kernel codes: 0 RGBA, 1 ASTC, 2 BC7, 3 BC1, 4 BC3, 5 ETC1, 6 ETC2 RGBA
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(U->hjobs.data(), U->jobs.p, sizeof(UastcJob) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  if (!outputs_on_device) {
    std::vector<UvolDnItem> dns; dns.reserve((size_t)n * L);
    // (a segment that failed on the device - status[] given - is downloaded like the others: its layers hold whatever the kernel wrote before it gave up)
    for (int s = 0; s < n; s++) for (uint32_t l = 0; l < L; l++) dns.push_back(UvolDnItem{ (uint8_t *)U->outs.p + layer_bytes * ((size_t)s * L + l), outp[(size_t)s * L + l], layer_bytes });
    const int rcd = uvol_download_staged(ctx, dns); if (rcd != UVOL_OK) return rcd;
  }
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->resolve_profile();
  if (status) { for (int s = 0; s < n; s++) status[s] = U->hjobs[s].status != 0 ? UVOL_E_ENCODE : UVOL_OK; return UVOL_OK; }
  for (int s = 0; s < n; s++) if (U->hjobs[s].status != 0) { ctx->set_error("segment %d: corrupt UASTC block or a mode this codec does not read (device status %d)", s, U->hjobs[s].status); return UVOL_E_ENCODE; }
  return UVOL_OK;
}
