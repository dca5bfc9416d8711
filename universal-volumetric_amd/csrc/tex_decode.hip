// tex_decode.hip — batched KTX2 / BasisLZ ETC1S -> RGBA8 decode (texture half).
// Replaces, for the RGBA32 target, the transcoder call the stock player makes per layer
// (reference src/lib/KTX2Loader.js:469-580 -> basis_transcoder `transcodeImage`); the bitstream is SURVEY Appendix B.0-B.4.
// Checked bit for bit in tests/ against a CPU restatement that is itself pinned on the reference's 50 .ktx2 fixtures.
//
// Structure (n segments per call, one launch per stage):
//   host              container header + supercompression global data offsets (a few hundred bytes per file)
//   k_tdec_tables     1 lane / segment: endpoint + selector codebooks (delta + Huffman), the four slice models
//   k_tdec_slices     1 wave / segment: lanes build the decode LUTs in LDS, lane 0 walks the slices' VLC streams
//                     (macroblock endpoint predictors + RLE, delta endpoints, selector history) -> per-block indices;
//                     P-frame skips copy the previous slice, so the slices of a segment are decoded in order
//   k_tdec_unpack     1 thread / block: (colour5, intensity table, 16 selectors) -> 16 RGBA8 texels, 16-byte row stores
// Bound: the VLC walk is serial per segment (latency / scalar ALU), the unpack is HBM-write bound (4 B per texel).
#include "uvol_common.hpp"

#define TD_MAX_LAYERS 64
#define TD_MAX_SYMS 16512          // >= TEX_MAX_CODEBOOK + history + RLE symbol
#define TD_LUT_EPM 10
#define TD_LUT_DEM 11
#define TD_LUT_SM 11
#define TD_LUT_RLE 8

struct DHuff { uint32_t n, valid; uint32_t first_code[18], first_idx[18], count[18]; uint32_t *sorted; uint8_t *sizes; };

struct TexDecJob {
  const uint8_t *file; uint32_t file_len;
  uint32_t width, height, layers, bx, by;
  uint32_t ashift, nsl;          // alpha slices: every image (layer) has two slices, colour 2l and alpha 2l + 1 (nsl = layers << ashift); a P-frame slice follows slice s - (1 << ashift)
  uint32_t ne, ns, ep_off, ep_len, sel_off, sel_len, tab_off, tab_len, level_off, level_len;
  uint32_t slice_flags[TD_MAX_LAYERS], slice_off[TD_MAX_LAYERS], slice_len[TD_MAX_LAYERS];
  uint8_t *endpoints;            // ne * 4: r5 g5 b5 inten
  uint32_t *selectors;           // ns: byte j = row j, texel x at bits 2x..2x+1
  uint16_t *ei, *si;             // nsl * bx * by
  DHuff hm[4];                   // epm, dem, sm, rle
  DHuff tmp[5];                  // codebook models (3 colour-delta, intensity-delta, selector byte-delta)
  uint32_t hist_size;
  uint8_t *out[TD_MAX_LAYERS];   // device RGBA8 buffers, width * height * 4 each, rows in stored order
  int32_t status;
};

// ---- LSB-first bit reader over global memory: aligned dwords, the next one always in flight ----
// (`pre` is requested through a formally divergent address and only moved to scalars when it is consumed, so the load
//  is really in flight while the previous 32 bits are decoded — see UVOL_LANE_ZERO / UVOL_READFIRST)
struct DBits {
  UVOL_G(const uint32_t) w; uint32_t nwords, wi, have, pre; int dz; unsigned long long win, consumed;
};
__device__ __forceinline__ void db_refill(DBits &B) {
  while (B.have <= 32) { B.win |= (unsigned long long)(uint32_t)UVOL_READFIRST(B.pre) << B.have; B.have += 32; B.wi++; B.pre = B.wi < B.nwords ? B.w[B.wi + B.dz] : 0u; }
}
__device__ __forceinline__ void db_init(DBits &B, const uint8_t *p, uint32_t nbytes) {
  const uint32_t a = (uint32_t)((size_t)p & 3);
  B.w = UVOL_TO_G(const uint32_t, reinterpret_cast<const uint32_t *>(p - a));
  B.nwords = (a + nbytes + 3) / 4; B.wi = 0; B.have = 0; B.win = 0; B.consumed = 0; B.dz = UVOL_LANE_ZERO();
  B.pre = B.nwords ? B.w[B.dz] : 0u;
  db_refill(B);
  B.win >>= 8 * a; B.have -= 8 * a;
}
__device__ __forceinline__ uint32_t db_peek(DBits &B, uint32_t n) { if (B.have < n) db_refill(B); return (uint32_t)(B.win & ((1ull << n) - 1)); }
__device__ __forceinline__ void db_skip(DBits &B, uint32_t n) { B.win >>= n; B.have -= n; B.consumed += n; }
__device__ __forceinline__ uint32_t db_get(DBits &B, uint32_t n) { const uint32_t v = db_peek(B, n); db_skip(B, n); return v; }

// ---- canonical (deflate-style) Huffman, max code length 16, codes matched MSB-first ----
__device__ inline int dh_init(DHuff &H, uint32_t n) {     // H.sizes[0..n) filled
  H.n = n; H.valid = 0;
  for (int l = 0; l < 18; l++) { H.count[l] = 0; H.first_code[l] = 0; H.first_idx[l] = 0; }
  for (uint32_t i = 0; i < n; i++) { const uint32_t s = H.sizes[i]; if (s > 16) return -1; if (s) H.count[s]++; }
  uint32_t code = 0, idx = 0;
  for (int l = 1; l <= 16; l++) { code = (code + H.count[l - 1]) << 1; H.first_code[l] = code; H.first_idx[l] = idx; idx += H.count[l]; }
  uint32_t fill[18]; for (int l = 0; l < 18; l++) fill[l] = H.first_idx[l];
  for (uint32_t i = 0; i < n; i++) { const uint32_t s = H.sizes[i]; if (s) H.sorted[fill[s]++] = i; }
  H.valid = idx > 0;
  return 0;
}
__device__ inline int dh_dec(const DHuff &H, DBits &B) {  // bit by bit (codebooks, and slice codes longer than the LUT)
  uint32_t code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | db_get(B, 1);
    const uint32_t c = H.count[l];
    if (c && code >= H.first_code[l] && code - H.first_code[l] < c) return (int)H.sorted[H.first_idx[l] + (code - H.first_code[l])];
  }
  return -1;
}
__device__ inline int d_read_huff(DBits &B, DHuff &out) {
  const int ZZ[21] = { 17, 18, 19, 20, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15, 16 };
  out.n = 0; out.valid = 0;
  const uint32_t total = db_get(B, 14);
  if (total == 0) return 0;
  if (total > TD_MAX_SYMS) return -1;
  const uint32_t ncl = db_get(B, 5); if (ncl < 1 || ncl > 21) return -1;
  uint8_t cls[21]; uint32_t csorted[21];
  for (int i = 0; i < 21; i++) cls[i] = 0;
  for (uint32_t i = 0; i < ncl; i++) cls[ZZ[i]] = (uint8_t)db_get(B, 3);
  DHuff clt; clt.sizes = cls; clt.sorted = csorted;
  if (dh_init(clt, 21)) return -1;
  uint32_t k = 0;
  while (k < total) {
    const int c = dh_dec(clt, B); if (c < 0) return -1;
    if (c <= 16) out.sizes[k++] = (uint8_t)c;
    else if (c == 17) { uint32_t r = 3 + db_get(B, 3); while (r-- && k < total + 200 && k < TD_MAX_SYMS + 256) out.sizes[k++] = 0; }
    else if (c == 18) { uint32_t r = 11 + db_get(B, 7); while (r-- && k < total + 200 && k < TD_MAX_SYMS + 256) out.sizes[k++] = 0; }
    else { uint32_t rep = c == 19 ? 3 + db_get(B, 2) : 7 + db_get(B, 7); if (k == 0) return -1; const uint8_t pv = out.sizes[k - 1]; while (rep-- && k < total + 200 && k < TD_MAX_SYMS + 256) out.sizes[k++] = pv; }
  }
  if (k != total) return -2;
  return dh_init(out, total);
}
__device__ __forceinline__ uint32_t d_vlc(DBits &B, int cb) {
  uint32_t v = 0; int ofs = 0;
  for (;;) { const uint32_t s = db_get(B, (uint32_t)cb + 1); v |= (s & ((1u << cb) - 1)) << ofs; ofs += cb; if (!(s >> cb) || ofs > 28) break; }
  return v;
}

// ---- K1: codebooks + slice models, one lane per segment ----
__global__ void __launch_bounds__(64) k_tdec_tables(TexDecJob *jobs) {
  TexDecJob &J = jobs[blockIdx.x];
  if (threadIdx.x != 0 || J.status != 0) return;
  const uint32_t ne = J.ne, ns = J.ns;
  { DBits R; db_init(R, J.file + J.ep_off, J.ep_len);
    if (d_read_huff(R, J.tmp[0]) || d_read_huff(R, J.tmp[1]) || d_read_huff(R, J.tmp[2]) || d_read_huff(R, J.tmp[3])) { J.status = -7; return; }
    const int gray = (int)db_get(R, 1);
    int prev[3] = { 16, 16, 16 }, pi = 0;
    for (uint32_t i = 0; i < ne; i++) {
      const int d = dh_dec(J.tmp[3], R); if (d < 0) { J.status = -7; return; } pi = (d + pi) & 7;
      for (int c = 0; c < (gray ? 1 : 3); c++) {
        const DHuff &mm = prev[c] <= 9 ? J.tmp[0] : (prev[c] <= 21 ? J.tmp[1] : J.tmp[2]);
        const int dd = dh_dec(mm, R); if (dd < 0) { J.status = -7; return; }
        prev[c] = (prev[c] + dd) & 31;
      }
      if (gray) prev[1] = prev[2] = prev[0];
      J.endpoints[4 * i] = (uint8_t)prev[0]; J.endpoints[4 * i + 1] = (uint8_t)prev[1]; J.endpoints[4 * i + 2] = (uint8_t)prev[2]; J.endpoints[4 * i + 3] = (uint8_t)pi;
    } }
  { DBits R; db_init(R, J.file + J.sel_off, J.sel_len);
    const int global = (int)db_get(R, 1), hybrid = (int)db_get(R, 1), raw = (int)db_get(R, 1);
    if (global || hybrid) { J.status = -8; return; }
    if (raw) { for (uint32_t i = 0; i < ns; i++) { uint32_t v = 0; for (int j = 0; j < 4; j++) v |= db_get(R, 8) << (8 * j); J.selectors[i] = v; } }
    else {
      if (d_read_huff(R, J.tmp[4])) { J.status = -8; return; }
      uint32_t prevb[4] = { 0, 0, 0, 0 };
      for (uint32_t i = 0; i < ns; i++) {
        uint32_t v = 0;
        for (int j = 0; j < 4; j++) {
          uint32_t cur;
          if (i == 0) cur = db_get(R, 8);
          else { const int d = dh_dec(J.tmp[4], R); if (d < 0) { J.status = -8; return; } cur = ((uint32_t)d ^ prevb[j]) & 255u; }
          prevb[j] = cur; v |= cur << (8 * j);
        }
        J.selectors[i] = v;
      }
    } }
  { DBits R; db_init(R, J.file + J.tab_off, J.tab_len);
    if (d_read_huff(R, J.hm[0]) || d_read_huff(R, J.hm[1]) || d_read_huff(R, J.hm[2]) || d_read_huff(R, J.hm[3])) { J.status = -9; return; }
    J.hist_size = db_get(R, 13); }
  if (!J.hm[0].valid || !J.hm[1].valid || !J.hm[2].valid || !J.hm[3].valid) { J.status = -9; return; }
  if (J.hist_size == 0 || J.hist_size > 64) { J.status = -10; return; }
}

// ---- K2: slices ----
// LUT entry: (symbol << 8) | code length, indexed by the next LB bits in stream order; 0 = code longer than LB bits
__device__ inline void lut_build(const DHuff &H, uint32_t *lut, int LB, uint32_t tid, uint32_t nthreads) {
  for (uint32_t k = tid; k < (1u << LB); k += nthreads) lut[k] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < H.n; i += nthreads) {
    const uint32_t l = H.sizes[i];
    if (!l || l > (uint32_t)LB) continue;
    // canonical code of symbol i: first_code[l] + (rank of i among the symbols of length l) = position in sorted[]
    uint32_t lo = H.first_idx[l], hi = lo + H.count[l];
    while (lo + 1 < hi) { const uint32_t mid = (lo + hi) / 2; if (H.sorted[mid] <= i) lo = mid; else hi = mid; }
    const uint32_t code = H.first_code[l] + (lo - H.first_idx[l]);
    uint32_t rc = 0; for (uint32_t b = 0; b < l; b++) rc |= ((code >> (l - 1 - b)) & 1u) << b;     // first bit read = MSB of the code
    for (uint32_t f = 0; f < (1u << ((uint32_t)LB - l)); f++) lut[rc | (f << l)] = (i << 8) | l;
  }
  __syncthreads();
}
template <int LB, typename LP>
__device__ __forceinline__ int lut_dec(const DHuff &H, LP lut, DBits &B) {
  const uint32_t e = lut[db_peek(B, LB)];
  if (e) { db_skip(B, e & 255u); return (int)(e >> 8); }
  return dh_dec(H, B);
}

// decoder state of one slice between rows (registers of the slice's lane 0)
struct SliceState { DBits R; uint32_t prev_sym, rep, prev_ei, sel_rle, rover; };
struct SliceLuts { UVOL_L(const uint32_t) epm; UVOL_L(const uint32_t) dem; UVOL_L(const uint32_t) sm; UVOL_L(const uint32_t) rle; };
// One row of blocks of one slice.  pe / pb: two rows of endpoint indices / macroblock predictor bits of THIS
// slice (LDS); pve / pvs: row y of the previous slice (for P-frame skips); oe / os: row y of this slice.  0 or an error code.
template <typename PV, typename OV>
__device__ __forceinline__ int tdec_row(const TexDecJob &J, const SliceLuts &T, SliceState &S, uint32_t y, bool is_p, UVOL_L(uint32_t) hist,
                                        UVOL_L(uint16_t) pe, UVOL_L(uint8_t) pb, uint32_t rowsz, PV pve, PV pvs, OV oe, OV os) {
  const uint32_t bx = J.bx, ne = J.ne, ns = J.ns, hs = J.hist_size, RLE = ns + hs;
  const uint32_t cur = y & 1;
  UVOL_L(uint16_t) pe_c = pe + cur * rowsz; UVOL_L(uint16_t) pe_o = pe + (cur ^ 1) * rowsz;
  UVOL_L(uint8_t) pb_c = pb + cur * rowsz; UVOL_L(uint8_t) pb_o = pb + (cur ^ 1) * rowsz;
  uint32_t pbits = 0;
  for (uint32_t x = 0; x < bx; x++) {
    if ((x & 1) == 0) {
      if ((y & 1) == 0) {
        if (S.rep) { S.rep--; pbits = S.prev_sym; }
        else {
          const int d = lut_dec<TD_LUT_EPM>(J.hm[0], T.epm, S.R); if (d < 0) return -12;
          if (d == 256) { S.rep = d_vlc(S.R, 4) + 3 - 1; pbits = S.prev_sym; } else { S.prev_sym = (uint32_t)d; pbits = (uint32_t)d; }
        }
        pb_o[x] = (uint8_t)(pbits >> 4);
      } else pbits = pb_c[x];
    }
    const uint32_t pred = pbits & 3; pbits >>= 2;
    uint32_t ei; bool skip = false;
    if (pred == 0) { if (x == 0) return -13; ei = S.prev_ei; }
    else if (pred == 1) { if (y == 0) return -13; ei = pe_o[x]; }
    else if (pred == 2) {
      if (is_p) { skip = true; ei = pve[x]; }
      else { if (x == 0 || y == 0) return -13; ei = pe_o[x - 1]; }
    } else { const int d = lut_dec<TD_LUT_DEM>(J.hm[1], T.dem, S.R); if (d < 0) return -12; ei = (uint32_t)d + S.prev_ei; if (ei >= ne) ei -= ne; }
    if (ei >= ne) return -14;
    pe_c[x] = (uint16_t)ei; S.prev_ei = ei;
    uint32_t si;
    if (skip) si = pvs[x];
    else {
      uint32_t sym;
      if (S.sel_rle > 0) { S.sel_rle--; sym = ns; }
      else {
        const int d = lut_dec<TD_LUT_SM>(J.hm[2], T.sm, S.R); if (d < 0) return -12; sym = (uint32_t)d;
        if (sym == RLE) { const int rr = lut_dec<TD_LUT_RLE>(J.hm[3], T.rle, S.R); if (rr < 0) return -12; S.sel_rle = rr == 63 ? d_vlc(S.R, 7) + 3 : (uint32_t)rr + 3; sym = ns; S.sel_rle--; }
      }
      if (sym >= ns) { const uint32_t h = sym - ns; if (h >= hs) return -14; si = hist[h]; if (h) { const uint32_t t = hist[h]; hist[h] = hist[h / 2]; hist[h / 2] = t; } }
      else { si = sym; hist[S.rover] = si; S.rover++; if (S.rover == hs) S.rover = hs / 2; }
    }
    if (si >= ns) return -14;
    oe[x] = (uint16_t)ei; os[x] = (uint16_t)si;
  }
  return 0;
}
__device__ __forceinline__ int tdec_slice_begin(const TexDecJob &J, uint32_t sl, SliceState &S, UVOL_L(uint32_t) hist, UVOL_L(uint16_t) pe, UVOL_L(uint8_t) pb, uint32_t rowsz) {
  if ((unsigned long long)J.slice_off[sl] + J.slice_len[sl] > J.level_len) return -11;
  if ((J.slice_flags[sl] & 2u) && (sl >> J.ashift) == 0) return -11;
  db_init(S.R, J.file + J.level_off + J.slice_off[sl], J.slice_len[sl]);
  for (uint32_t i = 0; i < J.hist_size; i++) hist[i] = i;
  S.rover = J.hist_size / 2; S.prev_sym = 0; S.rep = 0; S.prev_ei = 0; S.sel_rle = 0;
  for (uint32_t i = 0; i < 2 * rowsz; i++) pe[i] = 0;
  for (uint32_t i = 0; i < 2 * rowsz; i++) pb[i] = 0;
  return 0;
}

// LDS layout shared by both slice kernels: [4 LUTs][per slice-wave: hist 64 x u32 | pe 2 x rowsz u16 | pb 2 x rowsz u8 | row buffers ei, si: 2 x rowsz u16 each]
#define TD_LUT_WORDS ((1u << TD_LUT_EPM) + (1u << TD_LUT_DEM) + (1u << TD_LUT_SM) + (1u << TD_LUT_RLE))
// (row buffers: 2 deep, 4 deep with alpha slices - the consumer of a row is then two waves and two steps behind its producer)
__host__ __device__ inline uint32_t td_wave_words(uint32_t rowsz, uint32_t depth) { return 64 + rowsz /* pe: 2 rows of u16 */ + (2 * rowsz + 3) / 4 /* pb */ + depth * rowsz /* ei, si rows: depth x rowsz u16 each */; }

// Serial form (any layer count): one lane walks the slices of a segment one after the other; rows through global memory.
__global__ void __launch_bounds__(64) k_tdec_slices(TexDecJob *jobs) {
  TexDecJob &J = jobs[blockIdx.x];
  UVOL_DYN_SMEM(uint32_t, lds);
  const uint32_t lane = threadIdx.x;
  const bool ok = J.status == 0;
  uint32_t *l_epm = lds, *l_dem = l_epm + (1u << TD_LUT_EPM), *l_sm = l_dem + (1u << TD_LUT_DEM), *l_rle = l_sm + (1u << TD_LUT_SM);
  uint32_t *l_w = lds + TD_LUT_WORDS;
  const uint32_t bx = J.bx, by = J.by, rowsz = (bx + 2) & ~1u;
  if (ok) { lut_build(J.hm[0], l_epm, TD_LUT_EPM, lane, 64); lut_build(J.hm[1], l_dem, TD_LUT_DEM, lane, 64); lut_build(J.hm[2], l_sm, TD_LUT_SM, lane, 64); lut_build(J.hm[3], l_rle, TD_LUT_RLE, lane, 64); }
  if (!ok || lane != 0) return;
  SliceLuts T; T.epm = UVOL_TO_L(const uint32_t, l_epm); T.dem = UVOL_TO_L(const uint32_t, l_dem); T.sm = UVOL_TO_L(const uint32_t, l_sm); T.rle = UVOL_TO_L(const uint32_t, l_rle);
  UVOL_L(uint32_t) hist = UVOL_TO_L(uint32_t, l_w); UVOL_L(uint16_t) pe = UVOL_TO_L(uint16_t, reinterpret_cast<uint16_t *>(l_w + 64));
  UVOL_L(uint8_t) pb = UVOL_TO_L(uint8_t, reinterpret_cast<uint8_t *>(l_w + 64 + rowsz));
  const size_t nbk = (size_t)bx * by;
  const uint32_t st = 1u << J.ashift;
  for (uint32_t sl = 0; sl < J.nsl; sl++) {
    SliceState S;
    int rc = tdec_slice_begin(J, sl, S, hist, pe, pb, rowsz);
    const bool is_p = (J.slice_flags[sl] & 2u) != 0;
    const uint32_t pv = sl >= st ? sl - st : 0;            // the previous slice of the same kind
    for (uint32_t y = 0; y < by && !rc; y++) {
      const size_t ro = (size_t)y * bx;
      rc = tdec_row(J, T, S, y, is_p, hist, pe, pb, rowsz, UVOL_TO_G(const uint16_t, J.ei + pv * nbk + ro), UVOL_TO_G(const uint16_t, J.si + pv * nbk + ro),
                    UVOL_TO_G(uint16_t, J.ei + sl * nbk + ro), UVOL_TO_G(uint16_t, J.si + sl * nbk + ro));
    }
    if (!rc && S.R.consumed > 8ull * J.slice_len[sl]) rc = -15;
    if (rc) { J.status = rc; return; }
  }
}

// Pipelined form (<= 16 layers): one wave per slice in one workgroup.  A P-frame row only needs the SAME row of the
// previous slice, so wave s decodes row t - s at step t: a systolic pipeline with one workgroup barrier per row, rows handed
// from slice to slice through double-buffered LDS row buffers and written to global memory by all 64 lanes (coalesced).
__global__ void __launch_bounds__(1024) k_tdec_slices_pipe(TexDecJob *jobs) {
  TexDecJob &J = jobs[blockIdx.x];
  UVOL_DYN_SMEM(uint32_t, lds);
  __shared__ int s_fail;
  const uint32_t tid = threadIdx.x, lane = tid & 63, sl = tid >> 6, nthreads = blockDim.x;
  const bool ok = J.status == 0;
  uint32_t *l_epm = lds, *l_dem = l_epm + (1u << TD_LUT_EPM), *l_sm = l_dem + (1u << TD_LUT_DEM), *l_rle = l_sm + (1u << TD_LUT_SM);
  const uint32_t bx = J.bx, by = J.by, rowsz = (bx + 2) & ~1u, L = J.nsl, st = 1u << J.ashift, depth = 2u << J.ashift, ww = td_wave_words(rowsz, depth);
  if (tid == 0) s_fail = 0;
  if (ok) { lut_build(J.hm[0], l_epm, TD_LUT_EPM, tid, nthreads); lut_build(J.hm[1], l_dem, TD_LUT_DEM, tid, nthreads); lut_build(J.hm[2], l_sm, TD_LUT_SM, tid, nthreads); lut_build(J.hm[3], l_rle, TD_LUT_RLE, tid, nthreads); }
  if (!ok) return;                                   // uniform for the whole workgroup
  SliceLuts T; T.epm = UVOL_TO_L(const uint32_t, l_epm); T.dem = UVOL_TO_L(const uint32_t, l_dem); T.sm = UVOL_TO_L(const uint32_t, l_sm); T.rle = UVOL_TO_L(const uint32_t, l_rle);
  uint32_t *mine = lds + TD_LUT_WORDS + sl * ww, *prevw = lds + TD_LUT_WORDS + (sl >= st ? sl - st : 0) * ww;
  UVOL_L(uint32_t) hist = UVOL_TO_L(uint32_t, mine); UVOL_L(uint16_t) pe = UVOL_TO_L(uint16_t, reinterpret_cast<uint16_t *>(mine + 64));
  UVOL_L(uint8_t) pb = UVOL_TO_L(uint8_t, reinterpret_cast<uint8_t *>(mine + 64 + rowsz));
  const uint32_t rows_off = 64 + rowsz + (2 * rowsz + 3) / 4;                           // words; then ei[2][rowsz], si[2][rowsz] as u16
  UVOL_L(uint16_t) my_rows = UVOL_TO_L(uint16_t, reinterpret_cast<uint16_t *>(mine + rows_off));
  UVOL_L(const uint16_t) pv_rows = UVOL_TO_L(const uint16_t, reinterpret_cast<const uint16_t *>(prevw + rows_off));
  const size_t nbk = (size_t)bx * by;
  const bool is_p = sl < L && (J.slice_flags[sl] & 2u) != 0;
  SliceState S;
  if (sl < L && lane == 0) { const int rc = tdec_slice_begin(J, sl, S, hist, pe, pb, rowsz); if (rc) { s_fail = rc; } }
  __syncthreads();
  for (uint32_t t = 0; t < by + L - 1; t++) {
    const bool active = sl < L && t >= sl && t - sl < by;
    const uint32_t y = t - sl, buf = y & (depth - 1);
    if (active && lane == 0 && !s_fail) {
      const int rc = tdec_row(J, T, S, y, is_p, hist, pe, pb, rowsz, pv_rows + buf * rowsz, pv_rows + (depth + buf) * rowsz, my_rows + buf * rowsz, my_rows + (depth + buf) * rowsz);
      if (rc) s_fail = rc;
    }
    __syncthreads();
    if (active) {                                    // the finished row goes to global memory, 64 lanes wide
      uint16_t *ge = J.ei + sl * nbk + (size_t)y * bx, *gs = J.si + sl * nbk + (size_t)y * bx;
      for (uint32_t x = lane; x < bx; x += 64) { ge[x] = my_rows[buf * rowsz + x]; gs[x] = my_rows[(depth + buf) * rowsz + x]; }
    }
  }
  if (sl < L && lane == 0 && !s_fail && S.R.consumed > 8ull * J.slice_len[sl]) s_fail = -15;
  __syncthreads();
  if (tid == 0 && s_fail) J.status = s_fail;
}

// ---- K3: unpack, one thread per (block, layer, segment) ----
__global__ void __launch_bounds__(UVOL_BLOCK) k_tdec_unpack(TexDecJob *jobs) {
  TexDecJob &J = jobs[blockIdx.z];
  if (J.status != 0) return;
  const uint32_t layer = blockIdx.y, b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const uint32_t bx = J.bx, by = J.by, W = J.width, H = J.height;
  if (layer >= J.layers || b >= bx * by) return;
  const int INTEN[8][4] = { {-8, -2, 2, 8}, {-17, -5, 5, 17}, {-29, -9, 9, 29}, {-42, -13, 13, 42}, {-60, -18, 18, 60}, {-80, -24, 24, 80}, {-106, -33, 33, 106}, {-183, -47, 47, 183} };
  const uint32_t X = b % bx, Y = b / bx;
  const size_t o = (size_t)(layer << J.ashift) * bx * by + b;
  const uint8_t *e = J.endpoints + 4 * (size_t)J.ei[o]; const uint32_t sel = J.selectors[J.si[o]];
  int base[3]; for (int c = 0; c < 3; c++) base[c] = (e[c] << 3) | (e[c] >> 2);
  const int t = e[3];
  // alpha: the block of the image's alpha slice, green channel (what the basis transcoder takes)
  const uint8_t *ae = J.ashift ? J.endpoints + 4 * (size_t)J.ei[o + (size_t)bx * by] : e; const uint32_t asel = J.ashift ? J.selectors[J.si[o + (size_t)bx * by]] : 0u;
  const int abase = (ae[1] << 3) | (ae[1] >> 2), at = ae[3];
  uint8_t *out = J.out[layer];
  for (int y = 0; y < 4; y++) {
    const uint32_t py = Y * 4 + (uint32_t)y; if (py >= H) break;
    uint32_t px4[4];
    for (int x = 0; x < 4; x++) {
      const int d = INTEN[t][(sel >> (8 * y + 2 * x)) & 3];
      int r = base[0] + d, g = base[1] + d, bb = base[2] + d;
      r = r < 0 ? 0 : (r > 255 ? 255 : r); g = g < 0 ? 0 : (g > 255 ? 255 : g); bb = bb < 0 ? 0 : (bb > 255 ? 255 : bb);
      int a = 255;
      if (J.ashift) { a = abase + INTEN[at][(asel >> (8 * y + 2 * x)) & 3]; a = a < 0 ? 0 : (a > 255 ? 255 : a); }
      px4[x] = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)bb << 16) | ((uint32_t)a << 24);
    }
    uint32_t *row = reinterpret_cast<uint32_t *>(out + 4 * ((size_t)py * W + X * 4));
    if (X * 4 + 3 < W && (W & 3) == 0) *reinterpret_cast<uint4 *>(row) = make_uint4(px4[0], px4[1], px4[2], px4[3]);
    else for (int x = 0; x < 4; x++) if (X * 4 + (uint32_t)x < W) row[x] = px4[x];
  }
}

// ---- K3': ETC1 target (the `etc2` raw-texture family of the player, reference src/Interfaces.ts:19, src/V2/player.ts:338-356):
// every ETC1S block IS an ETC1 block — differential mode with a zero delta, both sub-blocks on the same intensity table —
// so the transcode is a re-pack of (colour5, table, selectors) into the 8-byte block, one thread per block.
// Block bytes: R5|dR3, G5|dG3, B5|dB3, table1(3)|table2(3)|diff(1)|flip(1), then pixel-index MSB and LSB planes (16 bits each,
// big-endian, pixel i = 4 * x + y); ETC1S selector 0..3 (dark -> bright) maps to the ETC1 index {3, 2, 0, 1}.
__global__ void __launch_bounds__(UVOL_BLOCK) k_tdec_etc1(TexDecJob *jobs) {
  TexDecJob &J = jobs[blockIdx.z];
  if (J.status != 0) return;
  const uint32_t layer = blockIdx.y, b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (layer >= J.layers || b >= J.bx * J.by) return;
  const size_t o = (size_t)layer * J.bx * J.by + b;
  const uint8_t *e = J.endpoints + 4 * (size_t)J.ei[o]; const uint32_t sel = J.selectors[J.si[o]];
  uint32_t msb = 0, lsb = 0;
  for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
    const uint32_t s2 = (sel >> (8 * y + 2 * x)) & 3u, idx = s2 == 0 ? 3u : (s2 == 1 ? 2u : (s2 == 2 ? 0u : 1u));
    const int i = 4 * x + y;
    msb |= (idx >> 1) << i; lsb |= (idx & 1u) << i;
  }
  const uint32_t t = e[3];
  uint8_t *out = J.out[layer] + 8 * (size_t)b;
  out[0] = (uint8_t)(e[0] << 3); out[1] = (uint8_t)(e[1] << 3); out[2] = (uint8_t)(e[2] << 3);
  out[3] = (uint8_t)((t << 5) | (t << 2) | 2u);
  out[4] = (uint8_t)(msb >> 8); out[5] = (uint8_t)msb; out[6] = (uint8_t)(lsb >> 8); out[7] = (uint8_t)lsb;
}

// ---- K3'b: ETC2 RGBA target (ETC2_EAC RGBA8: what the stock loader asks an ETC1S file WITH alpha for on ETC2 hardware,
// src/lib/KTX2Loader.js:672-676 transcoderFormat[1]).  16 bytes per block: an EAC alpha block, then the colour block - the same exact
// ETC1 re-pack as above (a differential block with zero deltas is a valid ETC2 block).  The alpha block of the alpha slice has four
// levels abase + {-a, -b, +b, +a} (green channel, clamped); EAC alpha is base + multiplier * table[j], eight levels out of 16 tables:
// every (table, multiplier 1..15) is tried with five base values around the middle of the levels, each level takes its nearest EAC
// level, the error is summed over the 16 pixels, the first best (table, multiplier, base ascending) wins.  Not a restatement of the
// basis transcoder's table-driven path (its tables are not in the reference): gated by alpha PSNR against the RGBA32 decode.
// Alpha block bytes: base, multiplier << 4 | table, then 16 x 3-bit indices, pixel i = 4 * x + y, first pixel in the top bits.
__device__ const int8_t EAC_MOD[16][8] = {
  { -3, -6, -9, -15, 2, 5, 8, 14 }, { -3, -7, -10, -13, 2, 6, 9, 12 }, { -2, -5, -8, -13, 1, 4, 7, 12 }, { -2, -4, -6, -13, 1, 3, 5, 12 },
  { -3, -6, -8, -12, 2, 5, 7, 11 }, { -3, -7, -9, -11, 2, 6, 8, 10 }, { -4, -7, -8, -11, 3, 6, 7, 10 }, { -3, -5, -8, -11, 2, 4, 7, 10 },
  { -2, -6, -8, -10, 1, 5, 7, 9 }, { -2, -5, -8, -10, 1, 4, 7, 9 }, { -2, -4, -8, -10, 1, 3, 7, 9 }, { -2, -5, -7, -10, 1, 4, 6, 9 },
  { -3, -4, -7, -10, 2, 3, 6, 9 }, { -1, -2, -3, -10, 0, 1, 2, 9 }, { -4, -6, -8, -9, 3, 5, 7, 8 }, { -3, -5, -7, -9, 2, 4, 6, 8 } };
__global__ void __launch_bounds__(UVOL_BLOCK) k_tdec_etc2a(TexDecJob *jobs) {
  TexDecJob &J = jobs[blockIdx.z];
  if (J.status != 0) return;
  const uint32_t layer = blockIdx.y, b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (layer >= J.layers || b >= J.bx * J.by) return;
  const size_t nbk = (size_t)J.bx * J.by, o = (size_t)(layer << J.ashift) * nbk + b;
  const int INTEN[8][4] = { {-8, -2, 2, 8}, {-17, -5, 5, 17}, {-29, -9, 9, 29}, {-42, -13, 13, 42}, {-60, -18, 18, 60}, {-80, -24, 24, 80}, {-106, -33, 33, 106}, {-183, -47, 47, 183} };
  uint8_t *out = J.out[layer] + 16 * (size_t)b;
  // ---- alpha half ----
  int al[4] = { 255, 255, 255, 255 }; uint32_t asel = 0, hist[4] = { 16, 0, 0, 0 };
  if (J.ashift) {
    const uint8_t *ae = J.endpoints + 4 * (size_t)J.ei[o + nbk]; asel = J.selectors[J.si[o + nbk]];
    for (int k = 0; k < 4; k++) { const int v = ((ae[1] << 3) | (ae[1] >> 2)) + INTEN[ae[3] & 7][k]; al[k] = v < 0 ? 0 : (v > 255 ? 255 : v); }
    hist[0] = 0; for (int i = 0; i < 16; i++) hist[(asel >> (2 * i)) & 3u]++;
  }
  int lo_l = 255, hi_l = 0;
  for (int k = 0; k < 4; k++) if (hist[k]) { lo_l = al[k] < lo_l ? al[k] : lo_l; hi_l = al[k] > hi_l ? al[k] : hi_l; }
  const int mid = (lo_l + hi_l + 1) >> 1;
  uint32_t best = 0xffffffffu; int bt = 0, bm = 1, bb = mid;
  for (int t = 0; t < 16 && best; t++) for (int m = 1; m < 16 && best; m++) for (int db = -2; db <= 2; db++) {
    const int base = mid + db; if (base < 0 || base > 255) continue;
    uint32_t err = 0;
    for (int k = 0; k < 4; k++) if (hist[k]) {
      int be = 1 << 30;
      for (int j = 0; j < 8; j++) { int v = base + m * EAC_MOD[t][j]; v = v < 0 ? 0 : (v > 255 ? 255 : v); const int d = v - al[k]; be = d * d < be ? d * d : be; }
      err += hist[k] * (uint32_t)be;
    }
    if (err < best) { best = err; bt = t; bm = m; bb = base; }
  }
  uint32_t lvl[4];
  for (int k = 0; k < 4; k++) { int be = 1 << 30; uint32_t bj = 0; for (int j = 0; j < 8; j++) { int v = bb + bm * EAC_MOD[bt][j]; v = v < 0 ? 0 : (v > 255 ? 255 : v); const int d = v - al[k]; if (d * d < be) { be = d * d; bj = (uint32_t)j; } } lvl[k] = bj; }
  unsigned long long bits = 0;
  for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { const int i = 4 * x + y; bits |= (unsigned long long)lvl[(asel >> (8 * y + 2 * x)) & 3u] << (45 - 3 * i); }
  out[0] = (uint8_t)bb; out[1] = (uint8_t)((bm << 4) | bt);
  for (int k = 0; k < 6; k++) out[2 + k] = (uint8_t)(bits >> (40 - 8 * k));
  // ---- colour half: the ETC1 re-pack ----
  const uint8_t *e = J.endpoints + 4 * (size_t)J.ei[o]; const uint32_t sel = J.selectors[J.si[o]];
  uint32_t msb = 0, lsb = 0;
  for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
    const uint32_t s2 = (sel >> (8 * y + 2 * x)) & 3u, idx = s2 == 0 ? 3u : (s2 == 1 ? 2u : (s2 == 2 ? 0u : 1u));
    const int i = 4 * x + y;
    msb |= (idx >> 1) << i; lsb |= (idx & 1u) << i;
  }
  const uint32_t t = e[3];
  out[8] = (uint8_t)(e[0] << 3); out[9] = (uint8_t)(e[1] << 3); out[10] = (uint8_t)(e[2] << 3);
  out[11] = (uint8_t)((t << 5) | (t << 2) | 2u);
  out[12] = (uint8_t)(msb >> 8); out[13] = (uint8_t)msb; out[14] = (uint8_t)(lsb >> 8); out[15] = (uint8_t)lsb;
}

// ---- K3'c: BC1 / BC3 targets (round 5; the stock loader's `dxtSupported` row, src/lib/KTX2Loader.js:610-618: TranscoderFormat.BC1 for an
// opaque file, BC3 for one with alpha slices - its choice where neither BPTC nor ETC2 exists).  Colour: the block's darkest / brightest ETC1S
// colour rounded to RGB565 are the endpoints of a four-colour BC1 block (palette c0, c1, (2 c0 + c1) / 3, (c0 + 2 c1) / 3: ETC1's inner
// colours sit at 0.31 - 0.39 of the span, BC1's at 1/3); each of the four ETC1S colours takes the palette entry nearest in squared error
// (clamping can bend the line), pixels follow their selectors.  An opaque BC1 block needs colour0 > colour1 as 16-bit numbers (otherwise
// index 3 means transparent black): endpoints are swapped and indices remapped where needed, equal endpoints use index 0 only.  BC3 = a BC4
// alpha block (alpha0 = highest, alpha1 = lowest level the block uses, eight-value mode, every level on its nearest palette value; equal
// levels: index 0) followed by the same colour block, which BC3 always reads in four-colour mode.  Re-fits, not restatements of the basis
// transcoder's tables (not in the reference): gated by PSNR against the RGBA32 decode through an independent decoder (tests/helpers.py).
// Layouts: colour0, colour1 (u16 LE, R in the top 5 bits), 16 x 2-bit indices raster order from bit 0; alpha0, alpha1, 16 x 3-bit indices.
template <bool BC3>
__global__ void __launch_bounds__(UVOL_BLOCK) k_tdec_bc13(TexDecJob *jobs) {
  TexDecJob &J = jobs[blockIdx.z];
  if (J.status != 0) return;
  const uint32_t layer = blockIdx.y, b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (layer >= J.layers || b >= J.bx * J.by) return;
  const size_t nbk = (size_t)J.bx * J.by, o = (size_t)(layer << J.ashift) * nbk + b;
  const int MODS[8][4] = { { -8, -2, 2, 8 }, { -17, -5, 5, 17 }, { -29, -9, 9, 29 }, { -42, -13, 13, 42 }, { -60, -18, 18, 60 }, { -80, -24, 24, 80 }, { -106, -33, 33, 106 }, { -183, -47, 47, 183 } };
  uint8_t *out = J.out[layer] + (BC3 ? 16 : 8) * (size_t)b;
  if (BC3) {
    int al[4] = { 255, 255, 255, 255 }; uint32_t asel = 0;
    if (J.ashift) {
      const uint8_t *ae = J.endpoints + 4 * (size_t)J.ei[o + nbk]; asel = J.selectors[J.si[o + nbk]];
      for (int k = 0; k < 4; k++) { const int v = ((ae[1] << 3) | (ae[1] >> 2)) + MODS[ae[3] & 7][k]; al[k] = v < 0 ? 0 : (v > 255 ? 255 : v); }
    }
    uint32_t used = J.ashift ? 0u : 1u;
    if (J.ashift) for (int i = 0; i < 16; i++) used |= 1u << ((asel >> (2 * i)) & 3u);
    int a0 = 0, a1 = 255;
    for (int k = 0; k < 4; k++) if ((used >> k) & 1u) { a0 = al[k] > a0 ? al[k] : a0; a1 = al[k] < a1 ? al[k] : a1; }
    uint32_t lvl[4] = { 0, 0, 0, 0 };
    if (a0 > a1) {
      for (int k = 0; k < 4; k++) {
        int be = 1 << 30;
        for (int j = 0; j < 8; j++) { const int v = j == 0 ? a0 : (j == 1 ? a1 : ((8 - j) * a0 + (j - 1) * a1) / 7); const int d = v - al[k]; if (d * d < be) { be = d * d; lvl[k] = (uint32_t)j; } }
      }
    }
    unsigned long long bits = 0;
    for (int i = 0; i < 16; i++) bits |= (unsigned long long)lvl[(asel >> (2 * i)) & 3u] << (3 * i);
    out[0] = (uint8_t)a0; out[1] = (uint8_t)a1;
    for (int k = 0; k < 6; k++) out[2 + k] = (uint8_t)(bits >> (8 * k));
    out += 8;
  }
  const uint8_t *e = J.endpoints + 4 * (size_t)J.ei[o]; const uint32_t sel = J.selectors[J.si[o]];
  int col[4][3];
  for (int c = 0; c < 3; c++) {
    const int base = (e[c] << 3) | (e[c] >> 2);
    for (int k = 0; k < 4; k++) { const int v = base + MODS[e[3] & 7][k]; col[k][c] = v < 0 ? 0 : (v > 255 ? 255 : v); }
  }
  // endpoints: brightest -> colour0, darkest -> colour1, rounded to 5 / 6 / 5 bits
  const int q0[3] = { (col[3][0] * 31 + 127) / 255, (col[3][1] * 63 + 127) / 255, (col[3][2] * 31 + 127) / 255 };
  const int q1[3] = { (col[0][0] * 31 + 127) / 255, (col[0][1] * 63 + 127) / 255, (col[0][2] * 31 + 127) / 255 };
  uint32_t c0 = (uint32_t)((q0[0] << 11) | (q0[1] << 5) | q0[2]), c1 = (uint32_t)((q1[0] << 11) | (q1[1] << 5) | q1[2]);
  int pal[4][3];
  { const int e0[3] = { (q0[0] << 3) | (q0[0] >> 2), (q0[1] << 2) | (q0[1] >> 4), (q0[2] << 3) | (q0[2] >> 2) };
    const int e1[3] = { (q1[0] << 3) | (q1[0] >> 2), (q1[1] << 2) | (q1[1] >> 4), (q1[2] << 3) | (q1[2] >> 2) };
    for (int c = 0; c < 3; c++) { pal[0][c] = e0[c]; pal[1][c] = e1[c]; pal[2][c] = (2 * e0[c] + e1[c]) / 3; pal[3][c] = (e0[c] + 2 * e1[c]) / 3; } }
  uint32_t map[4] = { 0, 0, 0, 0 };
  if (c0 != c1) {
    for (int k = 0; k < 4; k++) {
      int be = 1 << 30;
      for (int j = 0; j < 4; j++) { int d = 0; for (int c = 0; c < 3; c++) { const int t = pal[j][c] - col[k][c]; d += t * t; } if (d < be) { be = d; map[k] = (uint32_t)j; } }
    }
    if (c0 < c1) {                                          // four-colour mode wants colour0 > colour1: swap, 0 <-> 1 and 2 <-> 3
      const uint32_t t = c0; c0 = c1; c1 = t;
      for (int k = 0; k < 4; k++) map[k] ^= 1u;
    }
  }
  uint32_t idx = 0;
  for (int i = 0; i < 16; i++) idx |= map[(sel >> (2 * i)) & 3u] << (2 * i);
  out[0] = (uint8_t)c0; out[1] = (uint8_t)(c0 >> 8); out[2] = (uint8_t)c1; out[3] = (uint8_t)(c1 >> 8);
  out[4] = (uint8_t)idx; out[5] = (uint8_t)(idx >> 8); out[6] = (uint8_t)(idx >> 16); out[7] = (uint8_t)(idx >> 24);
}

// ---- K3'': BC7 target (what KTX2Loader picks on desktop GPUs, reference src/lib/KTX2Loader.js:591-689: astc, then bptc).  An ETC1S
// block has four colours base + {-a, -b, +b, +a}, clamped per channel.  Two single-subset BC7 modes can hold them with the
// darkest / brightest colour as endpoints: mode 5 (7-bit RGB endpoints widened by bit replication, 2-bit indices, weights
// 0/21/43/64 - close to ETC1's inner colours at ~0.35 of the span, and exact for black and white; separate 8-bit alpha = 255)
// and mode 6 (7-bit endpoints + p-bit, 4-bit indices; opaque alpha forces both p-bits to 1, so endpoints are odd values).
// Each of the four colours takes the index closest in squared error over the three channels (clamping can bend the line,
// hence the search); the block keeps the mode with the smaller error summed over its 16 pixels (ties: mode 5).  Not a
// restatement of the basis transcoder's table-driven mode-5 path: the gate for this target is PSNR against the RGBA32 decode.
// Bit layouts, LSB first.  Mode 5: 6 bits (1 << 5); rotation 2 (= 0); R0 R1 G0 G1 B0 B1, 7 bits each; A0 A1, 8 bits each;
// colour indices 1 + 15 * 2 bits; alpha indices 1 + 15 * 2 bits (all 0).  Mode 6: 7 bits (1 << 6); R0 R1 G0 G1 B0 B1 A0 A1,
// 7 bits each; P0 P1; indices 3 + 15 * 4 bits.  Pixel 0's index has its MSB implied 0: otherwise swap the endpoints and
// complement the indices.  Pixels in raster order.
__global__ void __launch_bounds__(UVOL_BLOCK) k_tdec_bc7(TexDecJob *jobs) {
  TexDecJob &J = jobs[blockIdx.z];
  if (J.status != 0) return;
  const uint32_t layer = blockIdx.y, b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (layer >= J.layers || b >= J.bx * J.by) return;
  const size_t o = (size_t)(layer << J.ashift) * J.bx * J.by + b;
  const uint8_t *e = J.endpoints + 4 * (size_t)J.ei[o]; const uint32_t sel = J.selectors[J.si[o]];
  const int MODS[8][4] = { { -8, -2, 2, 8 }, { -17, -5, 5, 17 }, { -29, -9, 9, 29 }, { -42, -13, 13, 42 }, { -60, -18, 18, 60 }, { -80, -24, 24, 80 }, { -106, -33, 33, 106 }, { -183, -47, 47, 183 } };
  const int W2[4] = { 0, 21, 43, 64 };
  const int W4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };
  int col[4][3], lo7[3], hi7[3];
  for (int c = 0; c < 3; c++) {
    const int base = (e[c] << 3) | (e[c] >> 2);
    for (int k = 0; k < 4; k++) { const int v = base + MODS[e[3] & 7][k]; col[k][c] = v < 0 ? 0 : (v > 255 ? 255 : v); }
    lo7[c] = col[0][c] >> 1; hi7[c] = col[3][c] >> 1;
  }
  uint32_t hist[4] = { 0, 0, 0, 0 };
  for (int i = 0; i < 16; i++) hist[(sel >> (2 * i)) & 3u]++;
  uint32_t idx5[4], idx6[4], err5 = 0, err6 = 0;
  for (int k = 0; k < 4; k++) {
    uint32_t b5 = 0, e5 = 0xffffffffu, b6 = 0, e6 = 0xffffffffu;
    for (uint32_t w = 0; w < 4; w++) {
      uint32_t err = 0;
      for (int c = 0; c < 3; c++) { const int l = (lo7[c] << 1) | (lo7[c] >> 6), h = (hi7[c] << 1) | (hi7[c] >> 6), d = ((l * (64 - W2[w]) + h * W2[w] + 32) >> 6) - col[k][c]; err += (uint32_t)(d * d); }
      if (err < e5) { e5 = err; b5 = w; }
    }
    for (uint32_t w = 0; w < 16; w++) {
      uint32_t err = 0;
      for (int c = 0; c < 3; c++) { const int d = (((2 * lo7[c] + 1) * (64 - W4[w]) + (2 * hi7[c] + 1) * W4[w] + 32) >> 6) - col[k][c]; err += (uint32_t)(d * d); }
      if (err < e6) { e6 = err; b6 = w; }
    }
    idx5[k] = b5; idx6[k] = b6; err5 += hist[k] * e5; err6 += hist[k] * e6;
  }
  unsigned long long lo, hi = 0; int pos;
  auto put = [&](unsigned long long v, int n) { if (pos < 64) { lo |= v << pos; if (pos + n > 64) hi |= v >> (64 - pos); } else hi |= v << (pos - 64); pos += n; };
  // A file with alpha slices (what the stock loader asks BC7 with alpha for, src/lib/KTX2Loader.js:672-676): always mode 5, whose alpha has
  // its own 8-bit endpoints and 2-bit indices.  The alpha block's four levels (green channel of the alpha slice, as the basis transcoder
  // takes them) give the endpoints (lowest / highest level, exact) and every level the nearest of the four interpolated values.
  uint32_t aidx[4] = { 0, 0, 0, 0 }, asel = 0; int a_lo = 255, a_hi = 255;
  if (J.ashift) {
    const size_t oa = o + (size_t)J.bx * J.by;
    const uint8_t *ae = J.endpoints + 4 * (size_t)J.ei[oa]; asel = J.selectors[J.si[oa]];
    int al[4];
    for (int k = 0; k < 4; k++) { const int v = ((ae[1] << 3) | (ae[1] >> 2)) + MODS[ae[3] & 7][k]; al[k] = v < 0 ? 0 : (v > 255 ? 255 : v); }
    a_lo = al[0]; a_hi = al[3];
    for (int k = 0; k < 4; k++) { uint32_t bw = 0; int be = 1 << 30; for (uint32_t w = 0; w < 4; w++) { const int d = ((a_lo * (64 - W2[w]) + a_hi * W2[w] + 32) >> 6) - al[k]; if (d * d < be) { be = d * d; bw = w; } } aidx[k] = bw; }
  }
  if (err5 <= err6 || J.ashift) {
    const bool swap = idx5[sel & 3u] >= 2u;              // pixel 0 = selector bits 0..1 (x = 0, y = 0)
    const bool aswap = aidx[asel & 3u] >= 2u;            // the alpha index set has its own anchor
    lo = 1ull << 5; pos = 8;
    for (int c = 0; c < 3; c++) { put((unsigned)(swap ? hi7[c] : lo7[c]), 7); put((unsigned)(swap ? lo7[c] : hi7[c]), 7); }
    put((unsigned)(aswap ? a_hi : a_lo), 8); put((unsigned)(aswap ? a_lo : a_hi), 8);
    for (int i = 0; i < 16; i++) { uint32_t ix = idx5[(sel >> (2 * i)) & 3u]; if (swap) ix = 3u - ix; put(ix, i == 0 ? 1 : 2); }
    if (J.ashift) for (int i = 0; i < 16; i++) { uint32_t ix = aidx[(asel >> (2 * i)) & 3u]; if (aswap) ix = 3u - ix; put(ix, i == 0 ? 1 : 2); }
  } else {
    const bool swap = idx6[sel & 3u] >= 8u;
    lo = 1ull << 6; pos = 7;
    for (int c = 0; c < 3; c++) { put((unsigned)(swap ? hi7[c] : lo7[c]), 7); put((unsigned)(swap ? lo7[c] : hi7[c]), 7); }
    put(127u, 7); put(127u, 7); put(1u, 1); put(1u, 1);
    for (int i = 0; i < 16; i++) { uint32_t ix = idx6[(sel >> (2 * i)) & 3u]; if (swap) ix = 15u - ix; put(ix, i == 0 ? 3 : 4); }
  }
  unsigned long long *out = (unsigned long long *)(J.out[layer] + 16 * (size_t)b);
  out[0] = lo; out[1] = hi;
}

// ================================================================================================
// host side
// ================================================================================================
struct TexDecState { uvol_devbuf files, slab, outs, jobs; std::vector<TexDecJob> hjobs; uint8_t *pinned = nullptr; size_t pinned_cap = 0; size_t max_lds = 64 * 1024; };

static inline uint32_t rd32h(const uint8_t *b) { uint32_t v; memcpy(&v, b, 4); return v; }
static inline uint64_t rd64h(const uint8_t *b) { uint64_t v; memcpy(&v, b, 8); return v; }

// container header + supercompression global data of one file
static int tdec_parse(const uint8_t *b, size_t n, TexDecJob &J) {
  static const uint8_t ident[12] = { 0xAB, 'K', 'T', 'X', ' ', '2', '0', 0xBB, '\r', '\n', 0x1A, '\n' };
  if (!b || n < 104 || n > 0xffffffffull || memcmp(b, ident, 12)) return -1;       // offsets are kept as uint32 below
  const uint32_t vk = rd32h(b + 12), W = rd32h(b + 20), H = rd32h(b + 24), layers = rd32h(b + 32), faces = rd32h(b + 36), levels = rd32h(b + 40), sc = rd32h(b + 44);
  const uint64_t sgd_off = rd64h(b + 64), sgd_len = rd64h(b + 72), lv_off = rd64h(b + 80), lv_len = rd64h(b + 88);
  if (vk != 0 || sc != 1 || levels != 1 || faces != 1 || W == 0 || H == 0 || W > 16384 || H > 16384) return -2;
  // untrusted 64-bit fields: compared without forming a sum that could wrap
  if (sgd_off > n || sgd_len > n - sgd_off || lv_off > n || lv_len > n - lv_off) return -3;
  const uint32_t nimg = layers ? layers : 1;
  if (nimg > TD_MAX_LAYERS) return -4;
  if (sgd_len < 20 + 20ull * nimg) return -5;
  const uint8_t *s = b + sgd_off;
  // alpha slices (basisu writes them for images with alpha; src/lib/KTX2Loader.js:493-497 reads them): the second offset / length pair
  // of every image desc, announced by a second DFD sample of channel 15 (AAA).  All images of a file have one or none.
  uint32_t any = 0, all = 1;
  for (uint32_t i = 0; i < nimg; i++) { if (rd32h(s + 20 + 20 * i + 16)) any = 1; else all = 0; }
  if (any != all) return -6;
  if (any) { const uint32_t dfd_off = rd32h(b + 48), dfd_len = rd32h(b + 52); if (dfd_len < 60 || dfd_off > n || dfd_len > n - dfd_off || (b[dfd_off + 44 + 3] & 15) != 15) return -6; }
  const uint32_t ash = any, nsl = nimg << ash;
  if (nsl > TD_MAX_LAYERS) return -4;
  J.width = W; J.height = H; J.layers = nimg; J.ashift = ash; J.nsl = nsl; J.bx = (W + 3) / 4; J.by = (H + 3) / 4;
  J.ne = (uint32_t)s[0] | ((uint32_t)s[1] << 8); J.ns = (uint32_t)s[2] | ((uint32_t)s[3] << 8);
  J.ep_len = rd32h(s + 4); J.sel_len = rd32h(s + 8); J.tab_len = rd32h(s + 12);
  for (uint32_t i = 0; i < nimg; i++) { const uint8_t *d = s + 20 + 20 * i;
    for (uint32_t k = 0; k <= ash; k++) { J.slice_flags[(i << ash) + k] = rd32h(d); J.slice_off[(i << ash) + k] = rd32h(d + 4 + 8 * k); J.slice_len[(i << ash) + k] = rd32h(d + 8 + 8 * k); } }
  const uint64_t p = sgd_off + 20 + 20ull * nimg;
  if (20 + 20ull * nimg + (uint64_t)J.ep_len + J.sel_len + J.tab_len > sgd_len) return -5;
  for (uint32_t i = 0; i < nsl; i++) if ((uint64_t)J.slice_off[i] + J.slice_len[i] > lv_len) return -6;
  J.ep_off = (uint32_t)p; J.sel_off = J.ep_off + J.ep_len; J.tab_off = J.sel_off + J.sel_len;
  J.level_off = (uint32_t)lv_off; J.level_len = (uint32_t)lv_len;
  if (J.ne == 0 || J.ns == 0 || J.ne > TD_MAX_SYMS - 80 || J.ns > TD_MAX_SYMS - 80) return -5;
  J.file_len = (uint32_t)n;
  return 0;
}

extern "C" int uvol_ktx2_info(const uint8_t *ktx2, size_t len, uint32_t *width, uint32_t *height, uint32_t *layers) {
  TexDecJob J; memset(&J, 0, sizeof J);
  { uint32_t w, h, l; uint64_t lo;                    // UASTC files of this codec (tex_uastc.hip)
    const int pr = uastc_ktx2_probe(ktx2, len, &w, &h, &l, &lo);
    if (pr == 0) { if (width) *width = w; if (height) *height = h; if (layers) *layers = l; return UVOL_OK; }
    if (pr == UASTC_PROBE_SUPERCOMPRESSED) {               // Zstandard-supercompressed UASTC: the size fields are in the header either way (nothing is inflated here)
      if (uastc_zstd_info(ktx2, len, &w, &h, &l) != 0) return UVOL_E_UNSUPPORTED;      // another scheme, or sizes that contradict each other
      if (width) *width = w; if (height) *height = h; if (layers) *layers = l; return UVOL_OK; } }
  if (tdec_parse(ktx2, len, J)) return UVOL_E_INVALID;
  if (width) *width = J.width; if (height) *height = J.height; if (layers) *layers = J.layers;
  return UVOL_OK;
}

int tdec_file_alpha(const uint8_t *b, size_t n) { TexDecJob J; memset(&J, 0, sizeof J); if (tdec_parse(b, n, J)) return -1; return J.ashift ? 1 : 0; }

int texdec_create(uvol_ctx *ctx) {
  ctx->texdec = new TexDecState();
#ifndef HIPEMU
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) == hipSuccess && v > 0) ctx->texdec->max_lds = (size_t)v;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tdec_slices_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->texdec->max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tdec_slices), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->texdec->max_lds);
  (void)hipGetLastError();
#else
  ctx->texdec->max_lds = 160 * 1024;
#endif
  return UVOL_OK;
}
void texdec_destroy(uvol_ctx *ctx) {
  TexDecState *t = ctx->texdec; if (!t) return;
  for (uvol_devbuf *b : { &t->files, &t->slab, &t->outs, &t->jobs }) if (b->p) (void)hipFree(b->p);
  if (t->pinned) (void)hipHostFree(t->pinned);
  delete t; ctx->texdec = nullptr;
}

#define DLAUNCH(k, grid, block, shmem, ...)                                                      \
  do {                                                                                           \
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch %s\n", #k); fflush(stderr); }              \
    hipLaunchKernelGGL(k, grid, block, shmem, ctx->stream, __VA_ARGS__);                         \
    if (uvol_debug()) { hipError_t e_ = hipStreamSynchronize(ctx->stream); if (e_ != hipSuccess) { fprintf(stderr, "[uvol] %s FAILED: %s\n", #k, hipGetErrorString(e_)); fflush(stderr); } } \
  } while (0)

// n segments (all of one width / height / layer count), rgba[s * layers + l] = width*height*4 bytes each
// status (optional): per-segment result codes; a segment whose stream turns out corrupt on the device then fails alone (UVOL_E_ENCODE in
// its slot, its layers are not written) and the call returns UVOL_OK - without it the first failing segment fails the call.  The
// container checks (parse, equal size / layer count) are the caller's with status[]: uvol_transcode_texture_segments_st sorts those out first.
int tex_decode_segments(uvol_ctx *ctx, const uint8_t *const *files, const size_t *lens, int n, uint8_t *const *rgba, size_t layer_cap, bool outputs_on_device, int target, int *status) {
  TexDecState *T = ctx->texdec;
  if (n <= 0) return UVOL_OK;
  T->hjobs.assign((size_t)n, TexDecJob{});
  size_t files_total = 0;
  std::vector<size_t> foff((size_t)n);
  for (int i = 0; i < n; i++) {
    const int rc = tdec_parse(files[i], lens[i], T->hjobs[i]);
    if (rc) { ctx->set_error("segment %d: not a KTX2 / BasisLZ ETC1S file this decoder supports (parse code %d)", i, rc); return rc == -2 || rc == -6 ? UVOL_E_UNSUPPORTED : UVOL_E_INVALID; }
    if (T->hjobs[i].width != T->hjobs[0].width || T->hjobs[i].height != T->hjobs[0].height || T->hjobs[i].layers != T->hjobs[0].layers || T->hjobs[i].ashift != T->hjobs[0].ashift) { ctx->set_error("segment %d: size / layer count differs from segment 0", i); return UVOL_E_INVALID; }
    foff[i] = files_total; files_total += (lens[i] + 16 + 255) & ~(size_t)255;
  }
  const TexDecJob &J0 = T->hjobs[0];
  const size_t nbk = (size_t)J0.bx * J0.by, L = J0.layers, NSL = J0.nsl;
  if (J0.ashift && target == 5) { ctx->set_error("segment 0 has alpha slices: BC1 is the opaque target (the stock loader asks such a file for BC3, src/lib/KTX2Loader.js:610-618: UVOL_TARGET_BC3)"); return UVOL_E_UNSUPPORTED; }
  if (J0.ashift && target == 1) { ctx->set_error("segment 0 has alpha slices: ETC1 is an opaque format (the stock loader asks such a file for ETC2 RGBA or BC7: uvol_transcode_texture_segments_etc2_rgba / _bc7)"); return UVOL_E_UNSUPPORTED; }
  const size_t layer_bytes = (target == 1 || target == 5) ? nbk * 8 : ((target == 2 || target == 4 || target == 6) ? nbk * 16 : (size_t)J0.width * J0.height * 4);   // 0: RGBA8, 1: ETC1 blocks, 2: BC7 blocks, 4: ETC2 RGBA blocks, 5: BC1 blocks, 6: BC3 blocks
  if (layer_cap < layer_bytes) { ctx->set_error("layer buffers too small: %zu < %zu", layer_cap, layer_bytes); return UVOL_E_NOSPACE; }
  // per-segment workspace: codebooks, block indices, Huffman size / sorted arrays of 9 models
  auto a256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t ws = a256((size_t)TD_MAX_SYMS * 4) + a256((size_t)TD_MAX_SYMS * 4) + 2 * a256(NSL * nbk * 2) + 9 * (a256((size_t)TD_MAX_SYMS + 512) + a256((size_t)TD_MAX_SYMS * 4 + 64));
  const size_t out_seg = outputs_on_device ? 0 : L * a256(layer_bytes);
  int rc;
  if ((rc = uvol_ensure(ctx, T->files, files_total + 64))) return rc;
  if ((rc = uvol_ensure(ctx, T->slab, ws * (size_t)n))) return rc;
  if ((rc = uvol_ensure(ctx, T->jobs, sizeof(TexDecJob) * (size_t)n))) return rc;
  if (!outputs_on_device && (rc = uvol_ensure(ctx, T->outs, out_seg * (size_t)n))) return rc;
  for (int i = 0; i < n; i++) {
    TexDecJob &J = T->hjobs[i];
    uint8_t *fd = (uint8_t *)T->files.p + foff[i];
    UVOL_HIP_CHECK(ctx, hipMemcpyAsync(fd, files[i], lens[i], hipMemcpyHostToDevice, ctx->stream));
    J.file = fd;
    uint8_t *w = (uint8_t *)T->slab.p + ws * (size_t)i; size_t o = 0;
    auto take = [&](size_t bytes) { uint8_t *p = w + o; o += a256(bytes); return p; };
    J.endpoints = take((size_t)TD_MAX_SYMS * 4); J.selectors = (uint32_t *)take((size_t)TD_MAX_SYMS * 4);
    J.ei = (uint16_t *)take(NSL * nbk * 2); J.si = (uint16_t *)take(NSL * nbk * 2);
    for (int k = 0; k < 9; k++) { DHuff &H = k < 4 ? J.hm[k] : J.tmp[k - 4]; H.sizes = take((size_t)TD_MAX_SYMS + 512); H.sorted = (uint32_t *)take((size_t)TD_MAX_SYMS * 4 + 64); }
    for (size_t l = 0; l < L; l++) J.out[l] = outputs_on_device ? rgba[(size_t)i * L + l] : (uint8_t *)T->outs.p + out_seg * (size_t)i + l * a256(layer_bytes);
    J.status = 0;
  }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(T->jobs.p, T->hjobs.data(), sizeof(TexDecJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  TexDecJob *dj = (TexDecJob *)T->jobs.p;
  const uint32_t rowsz = (J0.bx + 2) & ~1u;
  const uint32_t depth = 2u << J0.ashift;
  const size_t lds_serial = ((size_t)TD_LUT_WORDS + td_wave_words(rowsz, 2)) * 4, lds_pipe = ((size_t)TD_LUT_WORDS + NSL * td_wave_words(rowsz, depth)) * 4;
  const bool pipe = NSL <= 16 && lds_pipe <= T->max_lds;
  if ((pipe ? lds_pipe : lds_serial) > T->max_lds) { ctx->set_error("texture too wide for the slice decoder's row buffers"); return UVOL_E_UNSUPPORTED; }
  { uvol_ctx::Scope sc(ctx, "texdec.k1_tables", 0); DLAUNCH(k_tdec_tables, dim3((unsigned)n), dim3(64), 0, dj); }
  { uvol_ctx::Scope sc(ctx, "texdec.k2_slices", 0);
    if (pipe) DLAUNCH(k_tdec_slices_pipe, dim3((unsigned)n), dim3(64 * (unsigned)NSL), lds_pipe, dj);
    else DLAUNCH(k_tdec_slices, dim3((unsigned)n), dim3(64), lds_serial, dj); }
  { uvol_ctx::Scope sc(ctx, "texdec.k3_unpack", (uint64_t)n * L * layer_bytes);
    if (target == 1) DLAUNCH(k_tdec_etc1, dim3(uvol_blocks(nbk), (unsigned)L, (unsigned)n), dim3(UVOL_BLOCK), 0, dj);
    else if (target == 2) DLAUNCH(k_tdec_bc7, dim3(uvol_blocks(nbk), (unsigned)L, (unsigned)n), dim3(UVOL_BLOCK), 0, dj);
    else if (target == 4) DLAUNCH(k_tdec_etc2a, dim3(uvol_blocks(nbk), (unsigned)L, (unsigned)n), dim3(UVOL_BLOCK), 0, dj);
    else if (target == 5) DLAUNCH(k_tdec_bc13<false>, dim3(uvol_blocks(nbk), (unsigned)L, (unsigned)n), dim3(UVOL_BLOCK), 0, dj);
    else if (target == 6) DLAUNCH(k_tdec_bc13<true>, dim3(uvol_blocks(nbk), (unsigned)L, (unsigned)n), dim3(UVOL_BLOCK), 0, dj);
    else DLAUNCH(k_tdec_unpack, dim3(uvol_blocks(nbk), (unsigned)L, (unsigned)n), dim3(UVOL_BLOCK), 0, dj); }
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(T->hjobs.data(), dj, sizeof(TexDecJob) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n; i++) {
    if (status) status[i] = T->hjobs[i].status != 0 ? UVOL_E_ENCODE : UVOL_OK;
    if (T->hjobs[i].status != 0) { ctx->set_error("segment %d: corrupt BasisLZ stream (device status %d)", i, T->hjobs[i].status); if (!status) return UVOL_E_ENCODE; }
  }
  if (!outputs_on_device) {                                  // host outputs: one staged download (pinned double buffers, host threads copy out)
    std::vector<UvolDnItem> dns; dns.reserve((size_t)n * L);
    for (int i = 0; i < n; i++) if (T->hjobs[i].status == 0) for (size_t l = 0; l < L; l++) dns.push_back(UvolDnItem{ T->hjobs[i].out[l], rgba[(size_t)i * L + l], layer_bytes });
    const int rcd = uvol_download_staged(ctx, dns); if (rcd != UVOL_OK) return rcd;
  }
  ctx->resolve_profile();
  return UVOL_OK;
}
