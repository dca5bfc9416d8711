// geo_entropy.hpp - K7: rANS / rabs coders.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
// ------------------------------------------------------------------------------------------------
// K7: rANS (RAW scheme) — histogram (parallel), table build (serial, tiny), encode (serial per stream)
// ------------------------------------------------------------------------------------------------
// grid (blocks, stream, frame).  Alphabets that fit (<= HIST_LDS entries) are counted in LDS first: the six
// valence-context streams have 5 symbols, so global atomics would serialise on 5 addresses per frame.
#define HIST_LDS 2048          // 8 KiB: fits the LDS the resident walkers leave free; rarer, larger symbols go to global atomics
__global__ void __launch_bounds__(UVOL_BLOCK) k_hist(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.z];
  __shared__ uint32_t lh[HIST_LDS];
  const bool ok = J.status == 0;
  RansStream &S = J.rs[blockIdx.y];
  const uint32_t n = ok ? S.n : 0;
  const uint32_t nlds = S.alpha_cap < HIST_LDS ? S.alpha_cap : HIST_LDS;     // symbols below nlds are counted in LDS first
  const uint32_t per_block = 16 * UVOL_BLOCK, b0 = blockIdx.x * per_block;
  if (b0 >= n) return;                                   // block-uniform
  for (uint32_t k = threadIdx.x; k < nlds; k += UVOL_BLOCK) lh[k] = 0;
  __syncthreads();
  uint32_t mx = 0;
  for (uint32_t i = b0 + threadIdx.x; i < n && i < b0 + per_block; i += UVOL_BLOCK) {
    const uint32_t s = S.syms[i];
    if (s >= S.alpha_cap) { J.status = -30; continue; }
    if (s < nlds) atomicAdd(&lh[s], 1u); else atomicAdd(&S.freq[s], 1u);
    mx = s > mx ? s : mx;
  }
  for (int d = 32; d >= 1; d >>= 1) { uint32_t m2 = __shfl_xor(mx, d); mx = m2 > mx ? m2 : mx; }
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(&S.max_sym, mx);
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < nlds; k += UVOL_BLOCK) { const uint32_t v = lh[k]; if (v) atomicAdd(&S.freq[k], v); }
}

// RAnsSymbolEncoder::Create + table serialisation, one lane per stream
__global__ void __launch_bounds__(64) k_rans_tables(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  UVOL_SERIAL_PRIO();
  RansStream &S = J.rs[blockIdx.x];
  if (threadIdx.x != 0 || J.status != 0 || S.n == 0) return;
  const uint32_t ns = S.max_sym + 1;
  uint32_t uniq = 0; for (uint32_t i = 0; i < ns; i++) uniq += S.freq[i] != 0;
  int bl = 0; { uint32_t v = uniq; while (v) { bl++; v >>= 1; } } if (bl < 1) bl = 1;
  if (bl > 18) { J.status = -31; return; }
  int prec_bits = (3 * bl) / 2; prec_bits = prec_bits < 12 ? 12 : (prec_bits > 20 ? 20 : prec_bits);
  const uint32_t prec = 1u << prec_bits;
  S.prec_bits = (uint32_t)prec_bits;
  uint32_t *probs = S.probs;
  unsigned long long tot = 0; const double total = (double)S.n;
  for (uint32_t i = 0; i < ns; i++) {
    uint32_t p = 0;
    if (S.freq[i]) { p = (uint32_t)(((double)S.freq[i] / total) * (double)prec + 0.5); if (p == 0) p = 1; }
    probs[i] = p; tot += p;
  }
  if (tot != prec) {
    // stable ascending order of symbol ids by probability: counting sort on the probability value
    uint32_t *cnt = S.scratch, *ord = S.scratch + prec + 2;
    for (uint32_t v = 0; v <= prec + 1; v++) cnt[v] = 0;
    for (uint32_t i = 0; i < ns; i++) { uint32_t p = probs[i] > prec ? prec : probs[i]; cnt[p + 1]++; }
    for (uint32_t v = 1; v <= prec + 1; v++) cnt[v] += cnt[v - 1];
    for (uint32_t i = 0; i < ns; i++) { uint32_t p = probs[i] > prec ? prec : probs[i]; ord[cnt[p]++] = i; }
    if (tot < prec) probs[ord[ns - 1]] += (uint32_t)(prec - tot);
    else {
      long long err = (long long)tot - prec;
      while (err > 0) {
        const double rel = (double)prec / (double)tot;
        for (long long j = (long long)ns - 1; j > 0; j--) {
          const uint32_t sid = ord[j];
          if (probs[sid] <= 1) { if (j == (long long)ns - 1) err = 0; break; }
          int newp = (int)floor(rel * (double)probs[sid]);
          int fix = (int)probs[sid] - newp;
          if (fix == 0) fix = 1;
          if (fix >= (int)probs[sid]) fix = (int)probs[sid] - 1;
          if (fix > err) fix = (int)err;
          probs[sid] -= fix; tot -= fix; err -= fix;
          if (tot == prec) break;
        }
      }
    }
  }
  { uint32_t c = 0; for (uint32_t i = 0; i < ns; i++) { S.cum[i] = c; c += probs[i]; } }
  uint8_t *h = S.head; uint32_t o = 0;
  h[o++] = 1; h[o++] = (uint8_t)bl; o += g_put_varint(h + o, ns);
  for (uint32_t i = 0; i < ns;) {
    const uint32_t p = probs[i];
    if (p == 0) {
      uint32_t off = 0; while (off < 63 && i + off + 1 < ns && probs[i + off + 1] == 0) off++;
      h[o++] = (uint8_t)((off << 2) | 3); i += off + 1;
    } else {
      const int nb = p < (1u << 6) ? 0 : (p < (1u << 14) ? 1 : 2);
      h[o++] = (uint8_t)(((p << 2) | nb) & 0xff);
      for (int k = 0; k < nb; k++) h[o++] = (uint8_t)((p >> (8 * (k + 1) - 2)) & 0xff);
      i++;
    }
  }
  S.head_len = o;
}

// rANS (blockIdx.x < GEO_NSTREAM) and rabs (blockIdx.x >= GEO_NSTREAM) state machines, one wave per stream, all
// streams of all frames in ONE launch.  The state recurrence x' = (x / p) * prec + x % p + cum is the only serial part,
// so everything else is hoisted out of it: 64 symbols are fetched at a time (one per lane), every lane looks its own
// {prob, cum} up in the LDS table and derives an exact reciprocal of prob in parallel; the serial loop then only reads
// those back with v_readlane and runs on the scalar unit (s_mul_hi instead of a ~35-instruction integer division, no
// LDS access in the dependent chain).  Output bytes are staged one per lane and stored 64 at a time.
// Reciprocal (Alverson): for 2 <= d < 2^31, s = ceil(log2 d), m = ceil(2^(31+s) / d):  floor(x / d) = (x * m) >> (31 + s)
// for every x < 2^31 (error term x*e/(d*2^(31+s)) < 2^-s <= 1/d).  States here stay below 2^30 (Draco: x < 1024 * p).
__device__ __forceinline__ uint2 g_recip(uint32_t d) {          // {m, s - 1}; d == 1 yields x - 1 (callers compensate)
  if (d < 2) return make_uint2(0xffffffffu, 0u);               // (x * (2^32 - 1)) >> 32 = x - 1 for x >= 1
  const uint32_t sh = 32u - (uint32_t)__clz((int)(d - 1));
  const unsigned long long m = ((1ull << (31 + sh)) + d - 1) / d;
  return make_uint2((uint32_t)m, sh - 1);
}
// The table only needs to be close, not in the dependent chain: 1024 entries (8 KiB, static) keep every stream of every
// frame resident at once (14 one-wave workgroups per frame) and fit the LDS that resident walkers leave free; larger
// alphabets read {prob, cum} from global memory / L2.
#define RANS_LDS_ENTRIES 1024
// one rabs step with the constants of one bit value (LIM = 4096 * ls, MULT = 256 - ls)
#define RABS_STEP(LIM, M, SH, ADD, MULT)                                                                        \
  {                                                                                                             \
    if (st >= (LIM)) {                                                                                          \
      if (lane == (w & 63)) stage = st & 255;                                                                   \
      w++; st >>= 8;                                                                                            \
      if ((w & 63) == 0 && w <= cap) pay[w - 64 + lane] = (uint8_t)stage;                                       \
    }                                                                                                           \
    const uint32_t q_ = (uint32_t)(((unsigned long long)st * (M)) >> 32) >> (SH);                               \
    st = st + (ADD) + q_ * (MULT);                                                                              \
  }
__global__ void __launch_bounds__(64) k_entropy_encode(GeoJob *jobs, int dbg) {
  GeoJob &J = jobs[blockIdx.x];
#ifndef HIPEMU
  const unsigned long long t_begin = dbg ? wall_clock64() : 0ull;
#endif
  UVOL_SERIAL_PRIO();
  __shared__ uint2 tab[RANS_LDS_ENTRIES];
  const uint32_t lane = threadIdx.x;
  const bool ok = J.status == 0;
  uint32_t stage = 0, w = 0;
  if (blockIdx.y < GEO_NSTREAM) {
    RansStream &S = J.rs[blockIdx.y];
    const uint32_t n = ok ? S.n : 0;
    const uint32_t ns = S.max_sym + 1;
    const bool in_lds = ns <= RANS_LDS_ENTRIES;
    if (n && in_lds) for (uint32_t k = lane; k < ns; k += 64) tab[k] = make_uint2(S.probs[k], S.cum[k]);
    __syncthreads();
    if (!n) return;
    const uint32_t prec_bits = S.prec_bits, prec = 1u << prec_bits, L = prec * 4;
    const uint32_t *syms = S.syms;
    uint8_t *pay = S.pay + 8; const uint32_t cap = S.pay_cap - 80;
    uint32_t st = L;
    // software pipeline over chunks of 64 symbols (lane j = j-th symbol from the end of the remaining range):
    // symbols are fetched two chunks ahead, their table entries one chunk ahead, both overlapping the serial loop
#define RANS_LOAD_SY(H) (lane < (H) ? syms[(H) - 1 - lane] : 0u)
#define RANS_LOOKUP(SY) (in_lds ? tab[SY] : make_uint2(S.probs[SY], S.cum[SY]))
    uint32_t hi = n;
    uint32_t sy_nxt = RANS_LOAD_SY(hi);
    uint2 e_nxt = RANS_LOOKUP(sy_nxt);
    sy_nxt = RANS_LOAD_SY(hi > 64 ? hi - 64 : 0u);
    while (hi > 0) {
      const uint32_t cnt = hi < 64 ? hi : 64;
      const uint2 e = e_nxt;
      hi -= cnt;
      e_nxt = RANS_LOOKUP(sy_nxt);
      sy_nxt = RANS_LOAD_SY(hi > 64 ? hi - 64 : 0u);
      const uint2 rc = g_recip(e.x);
      const uint32_t ps = e.x | (rc.y << 24);                        // prob < 2^21, shift - 1 < 32
      const uint32_t cs = e.y + (e.x == 1 ? prec - 1 : 0);            // prob 1: the reciprocal yields x - 1, made up for here
      for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t pj = UVOL_READLANE(ps, j), p = pj & 0xffffffu, lim = 1024u * p;
        const uint32_t m = UVOL_READLANE(rc.x, j), cj = UVOL_READLANE(cs, j);
        if (st >= lim) {                                              // renormalise: k = bytes to emit (x < 2^30, lim >= 1024: at most 3)
          uint32_t k = 3u; k = (st >> 16) < lim ? 2u : k; k = (st >> 8) < lim ? 1u : k;
          const uint32_t pos = w & 63;
          if (__builtin_expect(pos + k >= 64, 0)) {                   // staging buffer wraps: byte by byte, flushing in between
            for (uint32_t i = 0; i < k; i++) {
              if (lane == (w & 63)) stage = st & 255;
              w++; st >>= 8;
              if ((w & 63) == 0 && w <= cap) pay[w - 64 + lane] = (uint8_t)stage;
            }
          } else {
            const uint32_t d = (lane - pos) & 63;
            if (d < k) stage = (st >> (8 * d)) & 255;
            w += k; st >>= 8 * k;
          }
        }
        const uint32_t q = (uint32_t)(((unsigned long long)st * m) >> 32) >> (pj >> 24);
        st = st + cj + q * (prec - p);                                // = q * prec + (st - q * p) + cum
      }
    }
#ifndef HIPEMU
    if (dbg && blockIdx.x == 0 && lane == 0) printf("[entropy] rans stream %d: n=%u alphabet=%u bytes=%u  %.3f ms\n", (int)blockIdx.y, n, ns, w, (double)(wall_clock64() - t_begin) * 1e-5);
#endif
    if (w + 4 > cap) { if (lane == 0) J.status = -32; return; }
    if (lane < (w & 63)) pay[(w & ~63u) + lane] = (uint8_t)stage;
    __threadfence_block();
    if (lane == 0) {
      st -= L;
      if (st < (1u << 6)) pay[w++] = (uint8_t)st;
      else if (st < (1u << 14)) { const uint32_t v = (1u << 14) + st; pay[w++] = v & 255; pay[w++] = (v >> 8) & 255; }
      else if (st < (1u << 22)) { const uint32_t v = (2u << 22) + st; pay[w++] = v & 255; pay[w++] = (v >> 8) & 255; pay[w++] = (v >> 16) & 255; }
      else { const uint32_t v = (3u << 30) + st; pay[w++] = v & 255; pay[w++] = (v >> 8) & 255; pay[w++] = (v >> 16) & 255; pay[w++] = (v >> 24) & 255; }
      const uint32_t vl = g_varint_len(w);
      g_put_varint(S.pay + 8 - vl, w);
      S.pay_off = 8 - vl; S.pay_len = vl + w;
    }
  } else {
    RabsStream &B = J.rb[blockIdx.y - GEO_NSTREAM];
    __syncthreads();
    if (!ok) return;
    const uint32_t n = B.n; const uint64_t total = n ? n : 1;
    const uint32_t p0raw = (uint32_t)(((double)B.zeros / (double)total) * 256.0 + 0.5);
    uint32_t p0 = p0raw < 255 ? p0raw : 255; if (p0 == 0) p0 = 1;
    p0 = UVOL_READLANE(p0, 0);
    const uint32_t p = 256 - p0;
    uint8_t *pay = B.buf + 8; const uint32_t cap = B.cap - 80;
    uint32_t st = 4096;
    // x' = (x / ls) * 256 + x % ls + add  =  x + add + q * (256 - ls);  ls == 1: q comes out as x - 1, compensated by 255
    const uint2 r1 = g_recip(p), r0 = g_recip(p0);
    const uint32_t m1 = UVOL_READLANE(r1.x, 0), s1 = UVOL_READLANE(r1.y, 0), m0 = UVOL_READLANE(r0.x, 0), s0 = UVOL_READLANE(r0.y, 0);
    const uint32_t a1 = (p == 1 ? 255u : 0u), a0 = p + (p0 == 1 ? 255u : 0u);
    const uint32_t lim1 = 4096u * p, lim0 = 4096u * p0, mu1 = 256u - p, mu0 = 256u - p0;
    uint32_t nxt = (lane < n && B.bits[n - 1 - lane] != 0) ? 1u : 0u;
    for (uint32_t hi = n; hi > 0;) {
      const uint32_t cnt = hi < 64 ? hi : 64;
      const unsigned long long bm = __ballot(nxt != 0);                // bit j = j-th bit from the end
      hi -= cnt;
      nxt = (lane < hi && B.bits[hi - 1 - lane] != 0) ? 1u : 0u;        // next chunk's read overlaps this chunk's serial loop
      for (uint32_t j = 0; j < cnt;) {                                  // runs of zeros in a tight loop with constant operands
        const unsigned long long rest = bm >> j;
        uint32_t run = rest ? (uint32_t)(__ffsll((long long)rest) - 1) : 64u; if (run > cnt - j) run = cnt - j;
        for (uint32_t r = 0; r < run; r++) RABS_STEP(lim0, m0, s0, a0, mu0);
        j += run;
        if (j < cnt) { RABS_STEP(lim1, m1, s1, a1, mu1); j++; }
      }
    }
#ifndef HIPEMU
    if (dbg && blockIdx.x == 0 && lane == 0) printf("[entropy] rabs stream %d: n=%u bytes=%u  %.3f ms\n", (int)blockIdx.y - GEO_NSTREAM, n, w, (double)(wall_clock64() - t_begin) * 1e-5);
#endif
    if (w + 3 > cap) { if (lane == 0) J.status = -33; return; }
    if (lane < (w & 63)) pay[(w & ~63u) + lane] = (uint8_t)stage;
    __threadfence_block();
    if (lane == 0) {
      st -= 4096;
      if (st < (1u << 6)) pay[w++] = (uint8_t)st;
      else if (st < (1u << 14)) { const uint32_t v = (1u << 14) + st; pay[w++] = v & 255; pay[w++] = (v >> 8) & 255; }
      else { const uint32_t v = (2u << 22) + st; pay[w++] = v & 255; pay[w++] = (v >> 8) & 255; pay[w++] = (v >> 16) & 255; }
      const uint32_t vl = g_varint_len(w);
      g_put_varint(B.buf + 8 - vl, w);
      B.buf[8 - vl - 1] = (uint8_t)p0;
      B.off = 8 - vl - 1; B.len = 1 + vl + w;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Lane-per-stream form of the entropy coder.  k_entropy_encode spends one wave per stream: 14 x frames waves, which at
// > 1000 frames per launch run in rounds.  Here every LANE encodes its own stream (the lanes of a wave take the same stream of
// consecutive frames, so their lengths are similar); nothing is in LDS.  k_rans_recip (parallel) turns the normalised
// probability table into one 16-byte entry per symbol {prob | shift << 24, cum (+ the prob == 1 correction), reciprocal}: a
// symbol costs one table load and a dozen integer instructions, symbols and entries are fetched four at a time one group
// ahead.  Output bytes are collected four to a word.  Byte-identical to k_entropy_encode.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(UVOL_BLOCK) k_rans_recip(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.z];
  if (J.status != 0) return;
  RansStream &S = J.rs[blockIdx.y];
  if (S.n == 0) return;
  const uint32_t k = blockIdx.x * UVOL_BLOCK + threadIdx.x, ns = S.max_sym + 1;
  if (k >= ns) return;
  const uint32_t p = S.probs[k], prec = 1u << S.prec_bits;
  const uint2 rc = g_recip(p);
  S.tab[k] = make_uint4(p | (rc.y << 24), S.cum[k] + (p == 1 ? prec - 1 : 0), rc.x, prec - p);
}
// 16-byte load through a typed global pointer (HIP's uint4 class cannot be read through an address-space-qualified pointer)
#ifdef HIPEMU
__device__ __forceinline__ uint4 g_ld4(const void *p) { return *reinterpret_cast<const uint4 *>(p); }
#else
typedef uint32_t uvol_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 g_ld4(UVOL_G(const void) p) { const uvol_u4 q = *(UVOL_G(const uvol_u4))p; return make_uint4(q.x, q.y, q.z, q.w); }
#endif
// Output of a lane coder.  On gfx950 loads and stores share vmcnt and the compiler has to wait for BOTH kinds (vmcnt(0)) whenever a
// store is outstanding next to a load it needs, so a loop that stores a few bytes per step and reads its next table entries drains
// its stores every iteration: ~2 us per group of 8 symbols, 240 ns per symbol, against ~50 ns of arithmetic (measured stream by
// stream).  The bytes are therefore staged in LDS (lgkmcnt, a different counter) and written out SB_FLUSH dwords at a time.
#define SB_STRIDE 65                       // dwords of LDS per lane (odd: lanes staging the same slot hit different banks)
#define SB_FLUSH 48                        // staged dwords that trigger a write-out at the next group boundary (a group adds <= 7)
struct SByteOut {
  UVOL_G(uint8_t) p; UVOL_L(uint32_t) stg; uint32_t w, cap, fill, nst; unsigned long long acc;
  __device__ __forceinline__ void init(uint8_t *dst, uint32_t cap_, uint32_t *lds_lane) { p = UVOL_TO_G(uint8_t, dst); stg = UVOL_TO_L(uint32_t, lds_lane); w = 0; cap = cap_; fill = 0; nst = 0; acc = 0; }
  // append the low k (0..3) bytes of v, least significant first
  __device__ __forceinline__ void put_n(uint32_t v, uint32_t k) {
    const uint32_t m = k == 0 ? 0u : (0xffffffffu >> (32 - 8 * k));
    acc |= (unsigned long long)(v & m) << (8 * fill);
    fill += k;
    if (fill >= 4) { stg[nst] = (uint32_t)acc; nst++; acc >>= 32; fill -= 4; }
  }
  __device__ __forceinline__ void write_out() {                        // staged dwords -> global memory; `w` = bytes written so far
    if (w + 4 * nst <= cap) for (uint32_t j = 0; j < nst; j++) *(UVOL_G(uint32_t))(p + w + 4 * j) = stg[j];
    w += 4 * nst; nst = 0;
  }
  __device__ __forceinline__ void group_end() { if (nst >= SB_FLUSH) write_out(); }
  __device__ __forceinline__ uint32_t bytes() const { return w + 4 * nst + fill; }
  __device__ __forceinline__ void flush() { write_out(); if (w + fill <= cap) for (uint32_t k = 0; k < fill; k++) p[w + k] = (uint8_t)(acc >> (8 * k)); }
};
__device__ inline void rans_encode_lane(GeoJob &J, RansStream &S, uint32_t *lds_lane) {
  const uint32_t n = S.n;
  if (!n) return;
  const uint32_t prec = 1u << S.prec_bits, L = prec * 4;
  UVOL_G(const uint32_t) syms = UVOL_TO_G(const uint32_t, S.syms); UVOL_G(const uint4) tab = UVOL_TO_G(const uint4, S.tab);
#define TAB(i) g_ld4(tab + (i))
#define SV(i) g_ld4(sv + (i))
  SByteOut O; O.init(S.pay + 8, S.pay_cap - 80, lds_lane);
  uint32_t st = L;
// one symbol: renormalise (at most three bytes leave: the state is below 2^(prec_bits + 10) <= 2^30, the limit at least 2^10) without
// a loop - the number of bytes is three compares, the bytes are the low bytes of the state -, then the exact-reciprocal update
#define SR_STEP(E)                                                                     \
  { const uint32_t p_ = (E).x & 0xffffffu, lim_ = p_ << 10;                             \
    uint32_t s_ = st;                                                                   \
    const bool c1_ = s_ >= lim_; s_ = c1_ ? s_ >> 8 : s_;                               \
    const bool c2_ = s_ >= lim_; s_ = c2_ ? s_ >> 8 : s_;                               \
    const bool c3_ = s_ >= lim_; s_ = c3_ ? s_ >> 8 : s_;                               \
    O.put_n(st, (uint32_t)c1_ + (uint32_t)c2_ + (uint32_t)c3_);                         \
    const uint32_t q_ = __umulhi(s_, (E).z) >> ((E).x >> 24);                           \
    st = s_ + (E).y + q_ * (E).w; }
  uint32_t hi = n;
  while (hi & 7u) { hi--; const uint4 e = TAB(syms[hi]); SR_STEP(e); O.group_end(); }          // the tail: the groups below are 32-byte aligned
  if (hi) {
    // software pipeline over groups of eight symbols: while group g is coded, the eight table entries of group g + 1 are in
    // flight (their symbols arrived an iteration earlier) and the symbols of group g + 2 are being fetched - a lane never issues
    // a load whose address it has to wait for, and an entry has ~8 symbol steps (> an L2 round trip) to arrive
    UVOL_G(const uint4) sv = (UVOL_G(const uint4))syms;
    uint4 s1a = SV(hi / 4 - 1), s1b = SV(hi / 4 - 2);                                        // symbols of the current group (high half first)
    uint4 s2a = s1a, s2b = s1b;
    if (hi >= 16) { s2a = SV(hi / 4 - 3); s2b = SV(hi / 4 - 4); }                            // ... of the next one
    uint4 e[8];
    e[0] = TAB(s1a.w); e[1] = TAB(s1a.z); e[2] = TAB(s1a.y); e[3] = TAB(s1a.x); e[4] = TAB(s1b.w); e[5] = TAB(s1b.z); e[6] = TAB(s1b.y); e[7] = TAB(s1b.x);
    while (hi) {
      hi -= 8;
      uint4 c[8];
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = e[k];
      if (hi) {
        e[0] = TAB(s2a.w); e[1] = TAB(s2a.z); e[2] = TAB(s2a.y); e[3] = TAB(s2a.x); e[4] = TAB(s2b.w); e[5] = TAB(s2b.z); e[6] = TAB(s2b.y); e[7] = TAB(s2b.x);
        if (hi >= 16) { s2a = SV(hi / 4 - 3); s2b = SV(hi / 4 - 4); }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) SR_STEP(c[k]);
      O.group_end();
    }
  }
#undef TAB
#undef SV
#undef SR_STEP
  uint32_t w = O.bytes();
  if (w + 4 > O.cap) { J.status = -32; return; }
  O.flush();
  uint8_t *pay = S.pay + 8;
  st -= L;
  if (st < (1u << 6)) pay[w++] = (uint8_t)st;
  else if (st < (1u << 14)) { const uint32_t v = (1u << 14) + st; pay[w++] = v & 255; pay[w++] = (v >> 8) & 255; }
  else if (st < (1u << 22)) { const uint32_t v = (2u << 22) + st; pay[w++] = v & 255; pay[w++] = (v >> 8) & 255; pay[w++] = (v >> 16) & 255; }
  else { const uint32_t v = (3u << 30) + st; pay[w++] = v & 255; pay[w++] = (v >> 8) & 255; pay[w++] = (v >> 16) & 255; pay[w++] = (v >> 24) & 255; }
  const uint32_t vl = g_varint_len(w);
  g_put_varint(S.pay + 8 - vl, w);
  S.pay_off = 8 - vl; S.pay_len = vl + w;
}
__device__ inline void rabs_encode_lane(GeoJob &J, RabsStream &B, uint32_t *lds_lane) {
  const uint32_t n = B.n; const uint64_t total = n ? n : 1;
  const uint32_t p0raw = (uint32_t)(((double)B.zeros / (double)total) * 256.0 + 0.5);
  uint32_t p0 = p0raw < 255 ? p0raw : 255; if (p0 == 0) p0 = 1;
  const uint32_t p = 256 - p0;
  SByteOut O; O.init(B.buf + 8, B.cap - 80, lds_lane);
  uint32_t st = 4096;
  const uint2 r1 = g_recip(p), r0 = g_recip(p0);
  const uint32_t a1 = (p == 1 ? 255u : 0u), a0 = p + (p0 == 1 ? 255u : 0u);
  const uint32_t lim1 = 4096u * p, lim0 = 4096u * p0, mu1 = 256u - p, mu0 = 256u - p0;
  UVOL_G(const uint8_t) bits = UVOL_TO_G(const uint8_t, B.bits);
#define SB_STEP(BYTE)                                                                  \
  { const bool one = (BYTE) != 0;                                                       \
    const uint32_t lim = one ? lim1 : lim0, m = one ? r1.x : r0.x, sh = one ? r1.y : r0.y, add = one ? a1 : a0, mul = one ? mu1 : mu0; \
    { const bool c_ = st >= lim; O.put_n(st, c_ ? 1u : 0u); st = c_ ? st >> 8 : st; }   \
    const uint32_t q = __umulhi(st, m) >> sh;                                           \
    st = st + add + q * mul; }
  // The flags are fetched 16 at a time, one chunk ahead: a byte load per step sits behind the coder's own stores (the compiler
  // cannot prove that they do not alias), i.e. one L2 round trip per bit - that, not the arithmetic, set the kernel's time.
  uint32_t i = n;
  while (i & 15u) { i--; SB_STEP(bits[i]); O.group_end(); }
  if (i) {
    UVOL_G(const uint4) bv = (UVOL_G(const uint4))bits;
    uint4 cur = g_ld4(bv + (i / 16 - 1)), nxt = cur;
    if (i >= 32) nxt = g_ld4(bv + (i / 16 - 2));
    while (i) {
      i -= 16;
      const uint4 c = cur; cur = nxt;
      if (i >= 32) nxt = g_ld4(bv + (i / 16 - 2));
      const uint32_t wv[4] = { c.w, c.z, c.y, c.x };
#pragma unroll
      for (int k = 0; k < 4; k++) { SB_STEP(wv[k] >> 24); SB_STEP((wv[k] >> 16) & 255u); SB_STEP((wv[k] >> 8) & 255u); SB_STEP(wv[k] & 255u); }
      O.group_end();
    }
  }
#undef SB_STEP
  uint32_t w = O.bytes();
  if (w + 3 > O.cap) { J.status = -33; return; }
  O.flush();
  uint8_t *pay = B.buf + 8;
  st -= 4096;
  if (st < (1u << 6)) pay[w++] = (uint8_t)st;
  else if (st < (1u << 14)) { const uint32_t v = (1u << 14) + st; pay[w++] = v & 255; pay[w++] = (v >> 8) & 255; }
  else { const uint32_t v = (2u << 22) + st; pay[w++] = v & 255; pay[w++] = (v >> 8) & 255; pay[w++] = (v >> 16) & 255; }
  const uint32_t vl = g_varint_len(w);
  g_put_varint(B.buf + 8 - vl, w);
  B.buf[8 - vl - 1] = (uint8_t)p0;
  B.off = 8 - vl - 1; B.len = 1 + vl + w;
}
// grid (frame blocks, stream); lanes of a wave = the same stream of W consecutive frames
__global__ void __launch_bounds__(64) k_entropy_simt(GeoJob *jobs, int n, int W) {
  UVOL_DYN_SMEM(uint32_t, lds);                                         // SB_STRIDE dwords per lane: the coders' output staging
  const int lane = (int)threadIdx.x;
  if (lane >= W) return;
  const int j = (int)blockIdx.x * W + lane;
  if (j >= n) return;
  GeoJob &J = jobs[j];
  if (J.status != 0) return;
  const int t = (int)blockIdx.y;
  if (t < GEO_NSTREAM) rans_encode_lane(J, J.rs[t], lds + lane * SB_STRIDE); else rabs_encode_lane(J, J.rb[t - GEO_NSTREAM], lds + lane * SB_STRIDE);
}

