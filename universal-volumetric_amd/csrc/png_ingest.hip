// png_ingest.hip — PNG scanlines un-filtered on the device (the ingest stage).
//
// `basisu` reads the PNGs itself (scripts/Encoder.py:274-292).  On the host a 2048^2 RGBA PNG costs ~35 core-ms: the zlib inflate and
// the un-filter pass (Sub / Up / Average / Paeth recurrences over 16.8 MB).  By default the inflate stays on the host (a serial bit stream per file,
// and the files of a batch inflate in parallel on the ingest threads; k_inflate below is the opt-in device form); the INFLATED scanlines - a filter-type byte + width * bpp filtered
// bytes per row - are uploaded as they are (the same 16.8 MB the un-filtered image would be) and un-filtered here, straight into the
// RGBA8 layers uvol_encode_texture_segments_dev reads.
//
// Average and Paeth are non-linear recurrences along a row AND depend on the row above, so a row cannot be split.  One WAVE per image:
// lane = channel * 16 + r, r = row of a band of 16 rows; row r runs ONE PIXEL behind row r - 1, so the pixel above (b) and above-left
// (c) are what the lane one below (same channel) produced one and two steps earlier - a DPP row shift, no LDS round trip -, the pixel to
// the left (a) is the lane's own last output.  A band takes width + 15 steps; the last row of a band stays in LDS for row 0 of the next.
// Every filter type runs the same straight-line step (the predictor is a select), so rows with different filters do not diverge.
// What a lone wave needs to run at ~0.1 us per step instead of 0.5 (tools/ingest_timing.py: it did not - 141 -> 134 ms per batch: the step is bound by instruction issue, ~1000 cycles for ~150 instructions of one wave): the filtered bytes
// of the next 16 steps are requested while the current 16 are processed (a global load per step was a memory round trip per step), and
// the outputs collect in an LDS ring (one byte write per lane and step) that the wave writes out 16 rows x 16 pixels at a time with
// 16-byte stores.
// Bit-identical to host/uvol_host.cpp read_png for 8-bit RGB / RGBA non-interlaced files (the other variants stay on the host).
#include "uvol_common.hpp"

struct PngJob { const uint8_t *raw; uint8_t *rgba; uint32_t w, h, ch; int32_t status; };

#define PNG_ROWS 16
__global__ void __launch_bounds__(64) k_png_unfilter(PngJob *jobs) {
  PngJob &J = jobs[blockIdx.x];
  UVOL_DYN_SMEM(uint32_t, lds);                            // [w] the band's last row (RGBA word per pixel), then [16 rows][32 pixels] output ring
  const uint32_t W = J.w, H = J.h, CH = J.ch; const size_t stride = (size_t)W * CH + 1;
  uint8_t *lastrow = reinterpret_cast<uint8_t *>(lds); uint8_t *ring = reinterpret_cast<uint8_t *>(lds + W);
  if (J.status != 0) return;                              // (the device inflate found the stream corrupt: block-uniform)
  const int lane = (int)threadIdx.x, r = lane & 15, c = lane >> 4;
  const bool chan = (uint32_t)c < CH;                     // (RGB files: the alpha lane only supplies 255)
  for (uint32_t x = (uint32_t)lane; x < W; x += 64) lds[x] = 0;
  __syncthreads();
  const uint32_t T = 16 * ((W + 15) / 16 + 1);            // steps per band: every chunk of 16 pixels of every row has left the ring by then
  for (uint32_t band = 0; band < H; band += PNG_ROWS) {
    const uint32_t row = band + (uint32_t)r; const bool live = row < H;
    const uint8_t *src = J.raw + stride * (size_t)(live ? row : 0) + 1 + c;
    const int ft = live ? (int)src[-1 - c] : 0;
    // (a filter-type byte above 4 is not PNG; read_png leaves such a row as it is, and so does the select below)
    uint32_t o1 = 0, o2 = 0;                              // this lane's outputs one / two steps ago (pixels x - 1, x - 2 of its row)
    uint32_t cur[4] = { 0, 0, 0, 0 }, nxt[4];
    // filtered bytes of steps [s0, s0 + 16) of this lane (pixel = step - r), packed four per word
#define PNG_FETCH(dst, s0)                                                                                         \
    do {                                                                                                           \
      _Pragma("unroll") for (int w_ = 0; w_ < 4; w_++) {                                                            \
        uint32_t v_ = 0;                                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++) {                                                          \
          const int x_ = (int)(s0) + 4 * w_ + j_ - r;                                                              \
          const uint32_t b_ = (live && chan && x_ >= 0 && x_ < (int)W) ? (uint32_t)src[(size_t)x_ * CH] : 0u;     \
          v_ |= b_ << (8 * j_);                                                                                    \
        }                                                                                                          \
        dst[w_] = v_;                                                                                              \
      }                                                                                                            \
    } while (0)
    PNG_FETCH(nxt, 0);
    for (uint32_t t = 0; t < T; t++) {
      const uint32_t j = t & 15u;
      if (j == 0) { cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2]; cur[3] = nxt[3]; PNG_FETCH(nxt, t + 16); }
      const uint32_t wsel = j < 8 ? (j < 4 ? cur[0] : cur[1]) : (j < 12 ? cur[2] : cur[3]);
      const uint32_t f = (wsel >> (8 * (j & 3u))) & 255u;
      const int x = (int)t - r;                           // this row's pixel at this step
      const bool on = live && x >= 0 && x < (int)W;
      // above / above-left: the lane one below, one / two steps ago; row 0 of the band reads the previous band's last row from LDS
      uint32_t up = UVOL_ROW_SHR1(o1), ul = UVOL_ROW_SHR1(o2);
      if (r == 0) {
        up = (on && band) ? (uint32_t)lastrow[4 * x + c] : 0u;
        ul = (on && band && x > 0) ? (uint32_t)lastrow[4 * (x - 1) + c] : 0u;
      }
      if (x <= 0) ul = 0;
      const int ia = x > 0 ? (int)o1 : 0, ib = (int)up, ic = (int)ul;
      // predictor by filter type: 0 none, 1 a, 2 b, 3 (a + b) / 2, 4 Paeth(a, b, c)
      const int pa = ib - ic < 0 ? ic - ib : ib - ic, pb = ia - ic < 0 ? ic - ia : ia - ic, pcv = ia + ib - 2 * ic, pc = pcv < 0 ? -pcv : pcv;
      const int paeth = (pa <= pb && pa <= pc) ? ia : (pb <= pc ? ib : ic);
      const int pred = ft == 1 ? ia : (ft == 2 ? ib : (ft == 3 ? ((ia + ib) >> 1) : (ft == 4 ? paeth : 0)));
      uint32_t out = (f + (uint32_t)pred) & 255u;
      if (!chan) out = 255u;                              // alpha of an RGB file
      if (!on) out = 0;
      if (on) {
        ring[(((uint32_t)r * 32u + ((uint32_t)x & 31u)) << 2) + (uint32_t)c] = (uint8_t)out;
        if (r == PNG_ROWS - 1 || row == H - 1) lastrow[4 * x + c] = (uint8_t)out;
      }
      o2 = o1; o1 = out;
      if (j == 15u) {
        // 16 rows x 16 pixels leave the ring: lane = row * 4 + group of four pixels; the chunk of row rr that is complete now
        UVOL_WAVE_SYNC();
        const int rr = lane >> 2, qd = lane & 3;
        const int k = (int)((t + 1 - (uint32_t)rr) >> 4) - 1;
        const uint32_t orow = band + (uint32_t)rr, x0 = (uint32_t)(16 * k + 4 * qd);
        if (k >= 0 && orow < H && x0 < W) {
          const uint32_t *rp = lds + W + (uint32_t)rr * 32u + (x0 & 31u);
          uint32_t *dst = reinterpret_cast<uint32_t *>(J.rgba) + (size_t)orow * W + x0;
          if (x0 + 3 < W && (W & 3u) == 0) *reinterpret_cast<uint4 *>(dst) = make_uint4(rp[0], rp[1], rp[2], rp[3]);
          else for (uint32_t q = 0; q < 4 && x0 + q < W; q++) dst[q] = rp[q];
        }
        UVOL_WAVE_SYNC();
      }
    }
#undef PNG_FETCH
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------------
// zlib inflate on the device: one WAVE per stream.  A DEFLATE stream is one serial chain of variable-length codes,
// so a stream cannot be split; what the wave has is 64 lanes for the parts that are parallel (table builds, match copies, the write-out)
// and an LDS that holds the whole 32 KiB window, so that no match ever waits for a global round trip: literals and matches go to a ring
// in LDS, the ring leaves for HBM 1 KiB at a time with 16-byte stores.  Every lane runs the same decode on the same (wave-uniform) bit
// buffer; Huffman symbols come from an 11-bit (literal / length) and a 9-bit (distance) look-up table in LDS, built per block by a
// canonical walk per table entry (every lane fills its entries, no replication loops of uneven length); longer codes take the canonical
// walk bit by bit.  Bit-identical to zlib for valid streams (tests: streams of every block type against the host zlib); corrupt streams
// end with a negative status and never read or write outside their buffers; the Adler-32 trailer is verified (sums by position, one
// reduction at the end: no serial chain).
// ------------------------------------------------------------------------------------------------
struct InflJob { const uint8_t *z; uint8_t *out; uint32_t zlen, out_cap, out_len; int32_t status; };
#define INF_WIN 32768u
#define INF_LBITS 11u
#define INF_DBITS 9u
#define INF_O_LUTL 32768u                               /* 2048 x u16 */
#define INF_O_LUTD (INF_O_LUTL + 4096u)                 /* 512 x u16 */
#define INF_O_LUTC (INF_O_LUTD + 1024u)                 /* 128 x u16 */
#define INF_O_SORTL (INF_O_LUTC + 256u)                 /* 288 x u16 */
#define INF_O_SORTD (INF_O_SORTL + 576u)                /* 32 x u16 */
#define INF_O_SORTC (INF_O_SORTD + 64u)                 /* 32 x u16 */
#define INF_O_CNT (INF_O_SORTC + 64u)                   /* 3 x 16 x u32 */
#define INF_O_LENS (INF_O_CNT + 192u)                   /* 320 + 32 code lengths */
#define INF_LDS (INF_O_LENS + 352u)
// canonical Huffman code of n symbols with code lengths lens[] (0 = unused): cnt[L] symbols per length, the symbols sorted by (length,
// symbol), and the table over the next `bits` stream bits (entry = symbol << 4 | length, 0 = a longer code or none).  Returns the
// "left" of the Kraft sum: < 0 over-subscribed, 0 complete, > 0 incomplete.  Block-wide (one wave), ends synchronised.
__device__ __forceinline__ int inf_build(const uint8_t *lens, uint32_t n, uint32_t *cnt, uint16_t *sorted, uint16_t *lut, uint32_t bits, int lane) {
  if (lane < 16) cnt[lane] = 0;
  __syncthreads();
  for (uint32_t s = (uint32_t)lane; s < n; s += 64) { const uint32_t L = lens[s]; if (L) atomicAdd(&cnt[L], 1u); }
  __syncthreads();
  int left = 1;
  for (int L = 1; L <= 15; L++) { left <<= 1; left -= (int)UVOL_READFIRST(cnt[L]); if (left < 0) break; }
  if (lane >= 1 && lane <= 15 && cnt[lane]) {           // lane L places the symbols of length L, in symbol order
    uint32_t k = 0; for (int L = 1; L < lane; L++) k += cnt[L];
    for (uint32_t s = 0; s < n; s++) if (lens[s] == (uint8_t)lane) sorted[k++] = (uint16_t)s;
  }
  __syncthreads();
  for (uint32_t e = (uint32_t)lane; e < (1u << bits); e += 64) {
    int code = 0, first = 0, index = 0; uint32_t ent = 0;
    for (uint32_t L = 1; L <= bits; L++) {
      code |= (int)((e >> (L - 1)) & 1u);
      const int c = (int)cnt[L];
      if (code - c < first) { ent = ((uint32_t)sorted[index + (code - first)] << 4) | L; break; }
      index += c; first += c; first <<= 1; code <<= 1;
    }
    lut[e] = (uint16_t)ent;
  }
  __syncthreads();
  return left;
}
// a code the table does not hold: the canonical walk over the next 15 bits (symbol, or -1: no such code)
__device__ __forceinline__ int inf_slow(unsigned long long bb, const uint32_t *cnt, const uint16_t *sorted, uint32_t &len) {
  int code = 0, first = 0, index = 0;
  for (int L = 1; L <= 15; L++) {
    code |= (int)((bb >> (L - 1)) & 1ull);
    const int c = (int)UVOL_READFIRST(cnt[L]);
    if (code - c < first) { len = (uint32_t)L; return (int)UVOL_READFIRST(sorted[index + (code - first)]); }
    index += c; first += c; first <<= 1; code <<= 1;
  }
  return -1;
}
// Adler-32 without a serial chain: with N bytes b[0..N), s1 = 1 + sum b, s2 = N + N * sum b - sum pos * b[pos] (mod 65521).  Every lane sums
// the bytes it writes out and their positions (64-bit, no intermediate modulo up to 2^29-byte outputs); one reduction at the end.
__device__ __forceinline__ void inf_adler16(const uint4 &v, uint32_t pos0, unsigned long long &A, unsigned long long &P) {
  const uint32_t w[4] = { v.x, v.y, v.z, v.w }; uint32_t a = 0, jb = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int j = 0; j < 4; j++) { const uint32_t b = (w[k] >> (8 * j)) & 255u; a += b; jb += (uint32_t)(4 * k + j) * b; }
  }
  A += a; P += (unsigned long long)pos0 * a + jb;
}
__global__ void __launch_bounds__(64) k_inflate(InflJob *jobs, PngJob *pj, uint32_t expect) {
  InflJob &J = jobs[blockIdx.x];
  UVOL_DYN_SMEM(uint8_t, lds);
  uint8_t *ring = lds;
  uint16_t *lutl = reinterpret_cast<uint16_t *>(lds + INF_O_LUTL), *lutd = reinterpret_cast<uint16_t *>(lds + INF_O_LUTD), *lutc = reinterpret_cast<uint16_t *>(lds + INF_O_LUTC);
  uint16_t *sortl = reinterpret_cast<uint16_t *>(lds + INF_O_SORTL), *sortd = reinterpret_cast<uint16_t *>(lds + INF_O_SORTD), *sortc = reinterpret_cast<uint16_t *>(lds + INF_O_SORTC);
  uint32_t *cntl = reinterpret_cast<uint32_t *>(lds + INF_O_CNT), *cntd = cntl + 16, *cntc = cntl + 32;
  uint8_t *lens = lds + INF_O_LENS, *clens = lens + 320;
  const int lane = (int)threadIdx.x;
  // every lane runs the same decode: what is read from memory is made wave-uniform explicitly (UVOL_READFIRST), so that the bit buffer, the
  // positions and every branch live in scalar registers (the compiler cannot know that an LDS read returns the same value in every lane)
  UVOL_G(const uint32_t) zw = UVOL_TO_G(const uint32_t, reinterpret_cast<const uint32_t *>(J.z));          // 16-byte aligned, >= 16 readable bytes behind the stream
  UVOL_G(const uint8_t) zb = UVOL_TO_G(const uint8_t, J.z);
  const uint32_t zlen = (uint32_t)UVOL_READFIRST(J.zlen), nw = (zlen + 3u) / 4u, cap = (uint32_t)UVOL_READFIRST(J.out_cap);
  UVOL_G(uint8_t) out = UVOL_TO_G(uint8_t, J.out); uint8_t *outg = J.out;
  const int dz = UVOL_LANE_ZERO();                      // (keeps the prefetched stream word in a vector register until it is consumed)
  unsigned long long bb = 0; uint32_t bc = 0, iw = 0;
  int err = 0;
  uint32_t opos = 0, flushed = 0;
  unsigned long long ad_a = 0, ad_p = 0;                // this lane's share of the Adler-32 sums
  uint32_t pend = 0, npend = 0;                         // literals not in the ring yet: lane k holds the k-th, they end at opos
#define INF_PEND_FLUSH() do { if (npend) { if ((uint32_t)lane < npend) ring[(opos - npend + (uint32_t)lane) & (INF_WIN - 1u)] = (uint8_t)pend; npend = 0; } } while (0)
  // words of the stream; behind its last byte the decoder sees ZEROS, whatever the device buffer holds there (the padding behind a stream is
  // never uploaded - a truncated stream decoded stale bytes of an earlier batch until a check caught it)
  const uint32_t tail_mask = (zlen & 3u) ? ((1u << (8u * (zlen & 3u))) - 1u) : 0xffffffffu;
#define INF_WORD(i_) ((i_) < nw ? (zw[(i_) + (uint32_t)dz] & ((i_) + 1u == nw ? tail_mask : 0xffffffffu)) : 0u)
#define INF_REFILL() do { if (bc <= 32u) { bb |= (unsigned long long)(uint32_t)UVOL_READFIRST(wnext) << bc; bc += 32u; iw++; wnext = INF_WORD(iw); } } while (0)
#define INF_BITS(n) ((uint32_t)(bb & ((1ull << (n)) - 1ull)))
#define INF_DROP(n) do { bb >>= (n); bc -= (n); } while (0)
  // complete KiB of the ring -> HBM (lane = 16 bytes); the ring never holds more than 1 KiB + one token that has not left
#define INF_FLUSH() do { while (opos - flushed >= 1024u) { UVOL_WAVE_SYNC(); \
      const uint4 v_ = *reinterpret_cast<const uint4 *>(ring + ((flushed + 16u * (uint32_t)lane) & (INF_WIN - 1u))); \
      *reinterpret_cast<uint4 *>(outg + flushed + 16u * (uint32_t)lane) = v_; inf_adler16(v_, flushed + 16u * (uint32_t)lane, ad_a, ad_p); flushed += 1024u; } } while (0)
  uint32_t wnext = INF_WORD(0u);
  INF_REFILL();
  { const uint32_t cmf = INF_BITS(8), flg = (uint32_t)(bb >> 8) & 255u; INF_DROP(16);
    if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u || (flg & 32u)) err = -1; }
  bool last = false;
  while (!last && !err) {
    __syncthreads();                                     // (the tables of the block before are not read any more)
    if (iw > nw + 1u) { err = -8; break; }               // the stream ended inside the block before (a run of empty non-final blocks terminates here)
    INF_REFILL();
    last = (bb & 1ull) != 0; const uint32_t type = (uint32_t)(bb >> 1) & 3u; INF_DROP(3);
    if (type == 0u) {
      INF_PEND_FLUSH();
      INF_DROP(bc & 7u); INF_REFILL();
      uint32_t len = INF_BITS(16); const uint32_t nlen = (uint32_t)(bb >> 16) & 0xffffu; INF_DROP(32);
      if ((len ^ nlen) != 0xffffu) { err = -3; break; }
      uint32_t p = 4u * iw - bc / 8u;                    // byte position of the stored data in the stream
      if ((unsigned long long)p + len > zlen) { err = -8; break; }
      if (opos + len > cap || opos + len < opos) { err = -7; break; }
      while (len) {
        const uint32_t n = len < 1024u ? len : 1024u;
        UVOL_WAVE_SYNC();
        for (uint32_t i = (uint32_t)lane; i < n; i += 64) ring[(opos + i) & (INF_WIN - 1u)] = zb[p + i];
        opos += n; p += n; len -= n;
        INF_FLUSH();
      }
      iw = p / 4u; bb = 0; bc = 0; wnext = INF_WORD(iw); INF_REFILL(); INF_DROP(8u * (p & 3u));
      continue;
    }
    if (type == 3u) { err = -2; break; }
    uint32_t hlit = 288, hdist = 30;
    if (type == 1u) {
      for (uint32_t s = (uint32_t)lane; s < 320u; s += 64) lens[s] = s < 144u ? 8 : (s < 256u ? 9 : (s < 280u ? 7 : (s < 288u ? 8 : 5)));
    } else {
      INF_REFILL();
      hlit = INF_BITS(5) + 257u; hdist = ((uint32_t)(bb >> 5) & 31u) + 1u; const uint32_t hclen = ((uint32_t)(bb >> 10) & 15u) + 4u; INF_DROP(14);
      if (hlit > 286u || hdist > 30u) { err = -4; break; }
      if (lane < 19) clens[lane] = 0;
      __syncthreads();
      for (uint32_t i = 0; i < hclen; i++) {
        if ((i & 7u) == 0u) INF_REFILL();
        const uint32_t v = INF_BITS(3); INF_DROP(3);
        // order of the code length code lengths: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
        const uint32_t o = i < 3u ? 16u + i : (i == 3u ? 0u : ((i & 1u) ? 8u - ((i - 3u) >> 1) : 8u + ((i - 4u) >> 1)));
        if (lane == 0) clens[o] = (uint8_t)v;
      }
      if (inf_build(clens, 19u, cntc, sortc, lutc, 7u, lane) < 0) { err = -4; break; }
      const uint32_t ntot = hlit + hdist; uint32_t idx = 0, prev = 0;
      while (idx < ntot) {
        INF_REFILL();
        const uint32_t e = (uint32_t)UVOL_READFIRST(lutc[bb & 127ull]);
        if (!e) { err = -4; break; }
        const uint32_t sym = e >> 4; INF_DROP(e & 15u);
        if (sym < 16u) { if (lane == 0) lens[idx] = (uint8_t)sym; idx++; prev = sym; continue; }
        uint32_t rep, val = 0;
        if (sym == 16u) { if (!idx) { err = -4; break; } val = prev; rep = 3u + INF_BITS(2); INF_DROP(2); }
        else if (sym == 17u) { rep = 3u + INF_BITS(3); INF_DROP(3); }
        else { rep = 11u + INF_BITS(7); INF_DROP(7); }
        if (idx + rep > ntot) { err = -4; break; }
        for (uint32_t k = (uint32_t)lane; k < rep; k += 64) lens[idx + k] = (uint8_t)val;
        idx += rep; prev = val;
      }
      if (err) break;
      __syncthreads();
      if (UVOL_READFIRST(lens[256]) == 0) { err = -4; break; }            // no end-of-block code
    }
    if (inf_build(lens, hlit, cntl, sortl, lutl, INF_LBITS, lane) < 0) { err = -4; break; }
    if (inf_build(lens + hlit, hdist, cntd, sortd, lutd, INF_DBITS, lane) < 0) { err = -4; break; }
    // One token per trip of the outer loop, after a run of literals taken in a tight inner loop: literals straight from the table while there
    // is `room` (neither the flush nor the capacity test can become due, the pending register has a free lane); they collect in a register,
    // lane k the k-th pending one, and reach the ring with one store per run (a store per literal queues in front of the next table
    // read).  The loops have few exits on purpose: every exit of a loop costs scalar bookkeeping in EVERY trip (the decode is bound by
    // instruction issue - profiles/r05_device_inflate.json), so the checks of a match are one test with one exit.
    for (;;) {
      uint32_t e, L, sym;
      { const uint32_t fast_end = flushed + 1024u < cap ? flushed + 1024u : cap, r1 = fast_end - opos, r2 = 64u - npend;
        const uint32_t k0 = npend, kend = npend + (r1 < r2 ? r1 : r2);
        for (;;) {
          INF_REFILL();
          e = (uint32_t)UVOL_READFIRST(lutl[bb & ((1ull << INF_LBITS) - 1ull)]);
          if (e - 1u >= 0xfffu || npend == kend) break;
          pend = lane == (int)npend ? (e >> 4) : pend; npend++; INF_DROP(e & 15u);
        }
        opos += npend - k0; }
      INF_PEND_FLUSH();
      if (e) { L = e & 15u; sym = e >> 4; }
      else { const int s_ = inf_slow(bb, cntl, sortl, L); if (s_ < 0) { err = -5; break; } sym = (uint32_t)s_; }
      INF_DROP(L);
      if (sym == 256u) break;
      if (sym < 256u) {                                    // (a literal the inner loop had no room for)
        if (opos >= cap) { err = -7; break; }
        ring[opos & (INF_WIN - 1u)] = (uint8_t)sym;
        opos++;
      } else {
        sym -= 257u;
        uint32_t len;
        if (sym < 8u) len = 3u + sym;
        else if (sym >= 28u) len = 258u;
        else { const uint32_t x = (sym - 4u) >> 2; len = 3u + ((4u + (sym & 3u)) << x) + INF_BITS(x); INF_DROP(x); }
        INF_REFILL();
        e = (uint32_t)UVOL_READFIRST(lutd[bb & ((1ull << INF_DBITS) - 1ull)]); uint32_t ds;
        if (e) { L = e & 15u; ds = e >> 4; }
        else { const int s_ = inf_slow(bb, cntd, sortd, L); if (s_ < 0) { err = -5; break; } ds = (uint32_t)s_; }
        INF_DROP(L);
        uint32_t dist;
        if (ds < 4u) dist = 1u + ds;
        else { const uint32_t x = ((ds >> 1) - 1u) & 15u; dist = 1u + ((2u + (ds & 1u)) << x) + INF_BITS(x); INF_DROP(x); }
        // length symbols 286 / 287 and distance symbols 30 / 31 do not exist; a distance cannot reach before the output, a match not beyond it
        if (sym > 28u || ds >= 30u || dist > opos || opos + len > cap) { err = -6; break; }
        // the copy: 64 bytes per round, reads before writes; a match that overlaps itself repeats its first `dist` bytes
        const uint32_t from = opos - dist;
        if (dist >= len) {                               // the usual case (PNG: the row above, the pixel to the left of a run): no modulo
          for (uint32_t b = 0; b < len; b += 64) {
            const uint32_t i = b + (uint32_t)lane; uint8_t v = 0;
            UVOL_WAVE_SYNC();
            if (i < len) v = ring[(from + i) & (INF_WIN - 1u)];
            UVOL_WAVE_SYNC();
            if (i < len) ring[(opos + i) & (INF_WIN - 1u)] = v;
          }
        } else {
          for (uint32_t b = 0; b < len; b += 64) {
            const uint32_t i = b + (uint32_t)lane; uint8_t v = 0;
            UVOL_WAVE_SYNC();
            if (i < len) v = ring[(from + i % dist) & (INF_WIN - 1u)];
            UVOL_WAVE_SYNC();
            if (i < len) ring[(opos + i) & (INF_WIN - 1u)] = v;
          }
        }
        opos += len;
      }
      INF_FLUSH();
      if (iw > nw + 1u) { err = -8; break; }
    }
  }
  // the trailer: Adler-32 of the output, big-endian, on the next byte boundary
  uint32_t want = 0;
  if (!err) { INF_DROP(bc & 7u); INF_REFILL(); const uint32_t t = INF_BITS(32); INF_DROP(32); want = (t >> 24) | ((t >> 8) & 0xff00u) | ((t << 8) & 0xff0000u) | (t << 24); }
  if (!err && 32ull * iw - bc > 8ull * zlen) err = -8;    // the stream ended inside a code or before its trailer
  INF_PEND_FLUSH();
  UVOL_WAVE_SYNC();
  for (uint32_t i = flushed + (uint32_t)lane; i < opos; i += 64) { const uint32_t b = ring[i & (INF_WIN - 1u)]; out[i] = (uint8_t)b; ad_a += b; ad_p += (unsigned long long)i * b; }
  __syncthreads();
  unsigned long long *red = reinterpret_cast<unsigned long long *>(lds + INF_O_LUTL);      // (the tables are done with)
  if (lane < 2) red[lane] = 0;
  __syncthreads();
  atomicAdd(&red[0], ad_a); atomicAdd(&red[1], ad_p % 65521ull);
  __syncthreads();
  if (!err) {
    const unsigned long long sa = red[0] % 65521ull, sp = red[1] % 65521ull, n = (unsigned long long)opos % 65521ull;
    const uint32_t s1 = (uint32_t)((1ull + sa) % 65521ull), s2 = (uint32_t)((n + n * sa + 65521ull - sp) % 65521ull);
    if (((s2 << 16) | s1) != want) err = -10;
  }
  if (lane == 0) {
    J.out_len = opos; J.status = err;
    if (pj) pj[blockIdx.x].status = err ? err : (opos != expect ? -20 : 0);
  }
#undef INF_REFILL
#undef INF_BITS
#undef INF_DROP
#undef INF_FLUSH
#undef INF_PEND_FLUSH
}
__global__ void __launch_bounds__(64) k_png_statuses(const PngJob *jobs, int32_t *st, int n) {
  const int i = (int)(blockIdx.x * 64 + threadIdx.x);
  if (i < n) st[i] = jobs[i].status;
}

// ================================================================================================
// host side
// ================================================================================================
struct PngState { uvol_devbuf raw, jobs; uvol_devbuf rgba[2]; std::vector<PngJob> hjobs; hipStream_t stream = nullptr; hipEvent_t done[2] = { nullptr, nullptr }; hipEvent_t up_done = nullptr; bool pending[2] = { false, false };
                  uvol_devbuf zin, ijobs, dstat; std::vector<InflJob> hij;                                      // device inflate: the compressed streams, their jobs, the statuses
                  int32_t *hstat[2] = { nullptr, nullptr }; size_t hstat_cap[2] = { 0, 0 }; int nstat[2] = { 0, 0 }; };   // per slot: the last call's per-image statuses (pinned)
int png_create(uvol_ctx *ctx) { ctx->png = new PngState(); return UVOL_OK; }
// the layers of the last un-filter call are ready before anything queued on `stream` from here on (the texture entry points call this
// before they read device inputs; uvol_sync too)
// (one event per slot: the encode of batch k waits for ITS slot, not for the un-filter of batch k + 1 queued meanwhile into the other one)
int png_order_before(uvol_ctx *ctx, hipStream_t stream, const uint8_t *const *layers, size_t n_layers) {
  PngState *S = ctx->png; if (!S) return UVOL_OK;
  // every layer of the call is looked at: a call may mix layers of both slots, or hold a pending slot's layer anywhere in its list
  bool need[2] = { false, false };
  for (int k = 0; k < 2; k++) {
    const uint8_t *b = (const uint8_t *)S->rgba[k].p;
    if (!S->pending[k] || !b) continue;
    for (size_t i = 0; i < n_layers && !need[k]; i++) need[k] = layers[i] >= b && layers[i] < b + S->rgba[k].cap;
  }
  for (int k = 0; k < 2; k++) if (need[k]) UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(stream, S->done[k], 0));
  return UVOL_OK;
}
int png_wait(uvol_ctx *ctx) {
  PngState *S = ctx->png; int rc = UVOL_OK;
  if (S) for (int k = 0; k < 2; k++) if (S->pending[k]) {
    const hipError_t e = hipEventSynchronize(S->done[k]); S->pending[k] = false;
    if (e != hipSuccess && rc == UVOL_OK) { ctx->set_error("hipEventSynchronize (png un-filter slot %d): %s", k, hipGetErrorString(e)); rc = UVOL_E_HIP; }
  }
  return rc;
}
void png_destroy(uvol_ctx *ctx) {
  PngState *S = ctx->png; if (!S) return;
  if (S->stream) { (void)hipStreamSynchronize(S->stream); (void)hipStreamDestroy(S->stream); }
  for (hipEvent_t e : S->done) if (e) (void)hipEventDestroy(e);
  if (S->up_done) (void)hipEventDestroy(S->up_done);
  for (uvol_devbuf *b : { &S->raw, &S->jobs, &S->rgba[0], &S->rgba[1], &S->zin, &S->ijobs, &S->dstat }) if (b->p) (void)hipFree(b->p);
  for (int32_t *h : S->hstat) if (h) (void)hipHostFree(h);
  delete S; ctx->png = nullptr;
}
// n images of one size: raw[i] = the INFLATED IDAT stream of an 8-bit non-interlaced RGB (channels 3) or RGBA (4) PNG, i.e. height rows
// of (1 filter-type byte + width * channels bytes), in host memory -> rgba_dev_out[i] = DEVICE pointer to width * height * 4 bytes,
// top row first, in the context's slot `slot` (valid until that slot is used again).  Runs on an ingest stream of its own and returns once
// the kernel is queued (the staged upload is what takes host time): the context's texture entry points order themselves behind it, so
// the un-filter of batch k + 1 runs beside the encode of batch k - one wave per image is 134 ms per batch whatever its size.
int png_unfilter_batch(uvol_ctx *ctx, const uint8_t *const *raw, int n, uint32_t w, uint32_t h, int channels, int slot, const uint8_t **rgba_dev_out) {
  return png_ingest_batch(ctx, raw, nullptr, n, w, h, channels, slot, rgba_dev_out);
}
// zlens != nullptr: raw[i] = the zlib stream itself (the concatenated IDAT chunks), zlens[i] bytes: inflated by k_inflate into the scanline
// buffer the un-filter kernel reads - the host neither inflates nor stages 16.8 MB per image, it uploads the 3 MB file.  Per-image
// statuses (corrupt stream, a size that is not height x (1 + width x channels)) come back through png_status().
int png_ingest_batch(uvol_ctx *ctx, const uint8_t *const *raw, const size_t *zlens, int n, uint32_t w, uint32_t h, int channels, int slot, const uint8_t **rgba_dev_out) {
  PngState *S = ctx->png;
  if (n <= 0) return UVOL_OK;
  if (slot < 0 || slot > 1 || (channels != 3 && channels != 4) || !w || !h || w > 8192 || h > 16384) { ctx->set_error("uvol_unfilter_png_batch_dev: slot 0 / 1, 3 or 4 channels, at most 8192 x 16384 (wider images: un-filter on the host)"); return UVOL_E_INVALID; }
  const size_t rbytes = ((size_t)w * channels + 1) * h, ra = (rbytes + 4 + 255) & ~(size_t)255, obytes = (size_t)w * h * 4;
  int rc;
  if (!S->stream) { if (uvol_make_stream(ctx, &S->stream) != hipSuccess || hipEventCreateWithFlags(&S->done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&S->done[1], hipEventDisableTiming) != hipSuccess) { ctx->set_error("ingest stream: creation failed"); return UVOL_E_HIP; } }
  hipStream_t saved = ctx->stream; ctx->stream = S->stream;            // (uvol_ensure, uvol_upload_staged, Scope use ctx->stream)
  struct Restore { uvol_ctx *c; hipStream_t s; ~Restore() { c->stream = s; } } restore_{ ctx, saved };
  if ((rc = uvol_ensure(ctx, S->raw, ra * (size_t)n))) return rc;
  if ((rc = uvol_ensure(ctx, S->rgba[slot], obytes * (size_t)n))) return rc;
  if ((rc = uvol_ensure(ctx, S->jobs, sizeof(PngJob) * (size_t)n))) return rc;
  // the per-image statuses' buffers first (a failing allocation after the kernels were queued returned without recording done[slot], and a
  // later encode of the slot's layers was not ordered behind them)
  S->nstat[slot] = 0;
  if ((rc = uvol_ensure(ctx, S->dstat, 4 * (size_t)n))) return rc;
  if (S->hstat_cap[slot] < (size_t)n) {
    if (S->hstat[slot]) { (void)hipEventSynchronize(S->done[slot]); (void)hipHostFree(S->hstat[slot]); S->hstat[slot] = nullptr; S->hstat_cap[slot] = 0; }
    const size_t c = ((size_t)n + 255) & ~(size_t)255;
    UVOL_HIP_CHECK(ctx, hipHostMalloc((void **)&S->hstat[slot], 4 * c, hipHostMallocDefault)); S->hstat_cap[slot] = c;
  }
  S->hjobs.assign((size_t)n, PngJob{});
  std::vector<UvolUpItem> ups; ups.reserve((size_t)n);
  for (int i = 0; i < n; i++) {
    if (!raw[i]) { ctx->set_error("PNG %d: no data", i); return UVOL_E_INVALID; }
    PngJob &J = S->hjobs[i]; J.raw = (const uint8_t *)S->raw.p + ra * (size_t)i; J.rgba = (uint8_t *)S->rgba[slot].p + obytes * (size_t)i; J.w = w; J.h = h; J.ch = (uint32_t)channels; J.status = 0;
    if (!zlens) ups.push_back(UvolUpItem{ ra * (size_t)i, raw[i], rbytes });
    rgba_dev_out[i] = J.rgba;
  }
  if (zlens) {
    size_t ztot = 0; std::vector<size_t> zoff((size_t)n);
    for (int i = 0; i < n; i++) { if (zlens[i] < 6 || zlens[i] > 0xfffffff0u) { ctx->set_error("PNG %d: not a zlib stream (%zu bytes)", i, zlens[i]); return UVOL_E_INVALID; } zoff[i] = ztot; ztot += (zlens[i] + 16 + 255) & ~(size_t)255; }
    if ((rc = uvol_ensure(ctx, S->zin, ztot))) return rc;
    if ((rc = uvol_ensure(ctx, S->ijobs, sizeof(InflJob) * (size_t)n))) return rc;
    S->hij.assign((size_t)n, InflJob{});
    for (int i = 0; i < n; i++) {
      InflJob &Z = S->hij[i]; Z.z = (const uint8_t *)S->zin.p + zoff[i]; Z.zlen = (uint32_t)zlens[i]; Z.out = (uint8_t *)S->raw.p + ra * (size_t)i; Z.out_cap = (uint32_t)rbytes; Z.out_len = 0; Z.status = 0;
      ups.push_back(UvolUpItem{ zoff[i], raw[i], zlens[i] });
    }
    { uvol_ctx::Scope sc(ctx, "ingest.png_upload", (uint64_t)ztot);
      if ((rc = uvol_upload_staged(ctx, (uint8_t *)S->zin.p, ups))) return rc; }
    UVOL_HIP_CHECK(ctx, hipMemcpyAsync(S->ijobs.p, S->hij.data(), sizeof(InflJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  } else {
    uvol_ctx::Scope sc(ctx, "ingest.png_upload", (uint64_t)rbytes * n);
    if ((rc = uvol_upload_staged(ctx, (uint8_t *)S->raw.p, ups))) return rc;
  }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(S->jobs.p, S->hjobs.data(), sizeof(PngJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  // "the host buffers may be re-used at once" holds for staged uploads (they have been copied into the pinned buffers) - inputs that all lie
  // in uvol_host_alloc memory go by DMA from where they lie, so the call waits for those copies (not for the kernels) before it returns
  bool direct = true; for (const UvolUpItem &it : ups) if (it.bytes && !uvol_host_pinned(it.src, it.bytes)) { direct = false; break; }
  if (direct) {
    if (!S->up_done) UVOL_HIP_CHECK(ctx, hipEventCreateWithFlags(&S->up_done, hipEventDisableTiming));
    UVOL_HIP_CHECK(ctx, hipEventRecord(S->up_done, ctx->stream));
  }
  if (zlens) {
    uvol_ctx::Scope sc(ctx, "ingest.png_inflate", (uint64_t)rbytes * n);
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch k_inflate\n"); fflush(stderr); }
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)n), dim3(64), (size_t)INF_LDS, ctx->stream, (InflJob *)S->ijobs.p, (PngJob *)S->jobs.p, (uint32_t)rbytes);
  }
  { uvol_ctx::Scope sc(ctx, "ingest.png_unfilter", (uint64_t)(rbytes + obytes) * n);
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch k_png_unfilter\n"); fflush(stderr); }
    hipLaunchKernelGGL(k_png_unfilter, dim3((unsigned)n), dim3(64), ((size_t)w + 16 * 32) * 4, ctx->stream, (PngJob *)S->jobs.p); }
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  // per-image statuses of this call, for png_status(): packed on the device, copied into the slot's pinned array
  hipLaunchKernelGGL(k_png_statuses, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const PngJob *)S->jobs.p, (int32_t *)S->dstat.p, n);
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(S->hstat[slot], S->dstat.p, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  S->nstat[slot] = n;
  UVOL_HIP_CHECK(ctx, hipEventRecord(S->done[slot], ctx->stream));
  S->pending[slot] = true;
  if (direct) UVOL_HIP_CHECK(ctx, hipEventSynchronize(S->up_done));      // (the kernels queued behind the copies run on)
  if (uvol_debug()) UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return UVOL_OK;
}
// statuses of the slot's last ingest call (waits for its kernels): UVOL_OK, or UVOL_E_INVALID for an image whose zlib stream is corrupt or
// does not hold height x (1 + width x channels) bytes; its layer is undefined, the other images of the call are not affected
int png_status(uvol_ctx *ctx, int slot, int *status, int n) {
  PngState *S = ctx->png;
  if (!S || slot < 0 || slot > 1 || n < 0 || n > S->nstat[slot]) { ctx->set_error("uvol_png_status: slot 0 / 1, at most the %d images of the slot's last call", S && slot >= 0 && slot <= 1 ? S->nstat[slot] : 0); return UVOL_E_INVALID; }
  if (S->done[slot]) { const hipError_t e = hipEventSynchronize(S->done[slot]); if (e != hipSuccess) { ctx->set_error("hipEventSynchronize (png slot %d): %s", slot, hipGetErrorString(e)); return UVOL_E_HIP; } }
  int worst = UVOL_OK;
  for (int i = 0; i < n; i++) {
    const int32_t d = S->hstat[slot][i]; const int st = d == 0 ? UVOL_OK : UVOL_E_INVALID;
    if (status) status[i] = st;
    if (st != UVOL_OK && worst == UVOL_OK) { worst = st; ctx->set_error("PNG %d: corrupt or unexpected zlib stream (device status %d)", i, d); }
  }
  return status ? UVOL_OK : worst;
}
