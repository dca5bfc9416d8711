// geom_encode.hip — hand-written HIP (gfx950) geometry encoder: OBJ-shaped arrays -> Draco 2.2 .drc.
//
// Replaces the arithmetic of HOT LOOP 1 of the reference (scripts/Encoder.py:256-267, one
// `draco_encoder -qp 11 -qt 10 -qn 8 -cl 7` process per frame) for a whole batch of frames at once.
// Kernel groups: K1 min/max+quantise, K2 value dedup, K3 corner table,
// K4 valence edgebreaker, K5 attribute DFS order, K6 prediction residuals, K7 rANS/rabs.
// Every kernel takes the device array of GeoJob and uses blockIdx.y (or .z) as the frame index, so
// a batch is ONE launch per stage: the parallel stages fill the chip, the serial walkers run one
// frame (or one entropy stream) per workgroup concurrently.
//
// There is no MFMA here by design: the path is integer/byte work bounded by dependent-load latency
// (walkers) and HBM/L2 bandwidth (parallel stages).
#include <chrono>
#include <thread>
#include "uvol_common.hpp"
#include "geom_device.hpp"
#include "uvol_ws.hpp"
#include <algorithm>
#include <map>

#define JOB_OR_RETURN GeoJob &J = jobs[blockIdx.y]; if (J.status != 0) return
// For kernels with barriers: the frame's status is read by ONE thread and the whole workgroup takes the same decision.  Another
// workgroup of the same frame may fail the frame at any moment; with a per-thread test some waves of a block would leave and
// the others wait for them at the barrier (the hardware tolerates that, the host emulation of tests/hipemu does not).
#define JOB_OR_RETURN_UNIFORM GeoJob &J = jobs[blockIdx.y]; { __shared__ int job_st_; if (threadIdx.x == 0) job_st_ = J.status; __syncthreads(); if (job_st_ != 0) return; }

// ------------------------------------------------------------------------------------------------
// block-level exclusive scan (wave shuffles + LDS), blockDim.x == UVOL_BLOCK
// ------------------------------------------------------------------------------------------------
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *total) {
  __shared__ uint32_t wsum[UVOL_BLOCK / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t x = v;
  for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int i = 0; i < UVOL_BLOCK / 64; i++) { if (i < w) base += wsum[i]; tot += wsum[i]; }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

// sum of v over the workgroup; every thread of the block calls it (no early returns before it)
__device__ inline uint32_t block_sum(uint32_t v) {
  __shared__ uint32_t acc;
  if (threadIdx.x == 0) acc = 0;
  __syncthreads();
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(&acc, v);
  __syncthreads();
  const uint32_t r = acc;
  __syncthreads();
  return r;
}

// scan selectors.  The producers of the KEEP / ELIG / EVENTS flags write the per-block sums themselves (block_sum), so only
// SCAN_ORI still runs k_scan_blocks; k_scan_sums turns the sums into block offsets for all four.
enum { SCAN_KEEP = 0, SCAN_ELIG = 1, SCAN_ORI = 2, SCAN_EVENTS = 3, SCAN_SEQ = 4 };      // SCAN_SEQ: per input corner (sequential connectivity)
// NOTE: written as value-returning selects on purpose.  The earlier form (out-references assigned in
// an if/else chain) was miscompiled by hipcc 7.2 -O3 for gfx950: the sel==2 arm left the pointer
// register undefined ("implicit-def $sgpr8_sgpr9" in the ISA) and the kernel faulted at address 0.
__device__ __forceinline__ const uint8_t *scan_flags(const GeoJob &J, int sel) { return sel == SCAN_SEQ ? J.sq_flag : (sel == SCAN_KEEP ? J.keep : (sel == SCAN_EVENTS ? J.evcnt : (sel == SCAN_ELIG ? J.elig : J.has_ori))); }
__device__ __forceinline__ uint32_t scan_count(const GeoJob &J, int sel) { return sel == SCAN_SEQ ? 3u * J.nf_in : (sel == SCAN_KEEP ? J.nf_in : (sel == SCAN_EVENTS ? J.nf : (sel == SCAN_ELIG ? J.nc : (J.has_uv ? J.ne_uv : 0u)))); }
__global__ void __launch_bounds__(UVOL_BLOCK) k_scan_blocks(GeoJob *jobs, int sel) {
  GeoJob &J = jobs[blockIdx.y];
  const uint8_t *flags = scan_flags(J, sel); const uint32_t n = scan_count(J, sel);
  if (blockIdx.x >= uvol_blocks_dev(n)) return;       // block-uniform exit
  uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  uint32_t v = (J.status == 0 && i < n) ? flags[i] : 0, tot;
  block_excl_scan(v, &tot);
  if (threadIdx.x == 0) (sel == SCAN_EVENTS ? J.bsum2 : J.bsum)[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_scan_sums(GeoJob *jobs, int sel) {
  GeoJob &J = jobs[blockIdx.y];
  const uint32_t nn = scan_count(J, sel);
  const uint32_t nblocks = uvol_blocks_dev(nn);
  uint32_t *bsum = sel == SCAN_EVENTS ? J.bsum2 : J.bsum;     // the event scan runs on the auxiliary stream
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nblocks; b0 += UVOL_BLOCK) {
    uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < nblocks ? bsum[i] : 0, tot;
    uint32_t ex = block_excl_scan(v, &tot);
    uint32_t c = carry;
    if (i < nblocks) bsum[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[nblocks] = carry;
}

// Corners per thread in the per-corner gather kernels (k_edge_match, k_aseg_a/b): they are latency-bound at full occupancy, so a
// thread issues every level of its dependent loads for GEO_ILP corners (one block stride apart: coalesced) before using any.
#define GEO_ILP 4
// ------------------------------------------------------------------------------------------------
// K2: bitwise value dedup.  table slot = (index+1), 0 = empty; final slot value = min index of the value.
// ------------------------------------------------------------------------------------------------
template <int NW>
__device__ inline bool words_eq(const uint32_t *a, const uint32_t *b) { bool e = true; for (int k = 0; k < NW; k++) e &= (a[k] == b[k]); return e; }

template <int NW>
__global__ void __launch_bounds__(UVOL_BLOCK) k_dedup(GeoJob *jobs, int which, int phase) {
  JOB_OR_RETURN;
  const uint32_t n = which == 0 ? J.n_pos : (which == 1 ? J.n_uv : J.n_nrm);
  const uint32_t *data = (const uint32_t *)(which == 0 ? J.pos : (which == 1 ? J.uv : J.nrm));
  uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (i >= n || data == nullptr) return;
  // phase 1 without duplicates (the usual case: phase 0 found no two equal values) is the identity: no second round of probes
  if (phase == 1 && J.n_dup[which] == 0) { J.canon[which][i] = i; return; }
  uint32_t *tab = J.dd_tab[which]; const uint32_t cap = J.dd_cap[which];
  uint32_t w[NW]; uint64_t h = 1469598103934665603ULL;
  for (int k = 0; k < NW; k++) { w[k] = data[(size_t)i * NW + k]; h = g_mix64(h ^ w[k]); }
  uint32_t s = (uint32_t)h & (cap - 1);
  for (uint32_t guard = 0; guard <= cap; guard++) {
    uint32_t cur = tab[s];
    if (phase == 0 && cur == 0) { uint32_t old = atomicCAS(&tab[s], 0u, i + 1); if (old == 0) return; cur = old; }
    if (cur == 0) break;
    if (words_eq<NW>(w, data + (size_t)(cur - 1) * NW)) {
      if (phase == 0) { atomicMin(&tab[s], i + 1); J.n_dup[which] = 1; } else J.canon[which][i] = cur - 1;
      return;
    }
    s = (s + 1) & (cap - 1);
  }
  if (phase == 1) J.status = -20;
}

// ------------------------------------------------------------------------------------------------
// K2, partitioned form (the default).  The hash table above costs one device-scope atomic on a random 64-byte line per
// value: memory-side read-modify-writes that do not cache (47 MB of HBM traffic per 100 k-vertex frame for 3.2 MB of values,
// profiles/r02_n).  Here the values are first partitioned by the top bits of their hash (count -> scan -> scatter of 16-byte
// {index, words} records: streaming passes, the only atomics are LDS counters), then every bin (~1 k values) is resolved by
// ONE workgroup in an LDS hash table.  canon[] = lowest index among bitwise-equal values, exactly as before.  A bin with more
// distinct values than the table holds (hash skew) fails the frame with GEO_E_DD_OVERFLOW; the host re-encodes it with the
// hash-table kernels.  grid z = attribute (0 pos, 1 uv, 2 normals), y = frame.
// ------------------------------------------------------------------------------------------------
#define DD_TILE 1024                         // values per workgroup in the count / scatter passes
#define DD_MAXBINS 1024
#define DD_SLOTS 4096                        // LDS hash slots per bin
struct DdSrc { const uint32_t *data; uint32_t n, nw; };
__device__ __forceinline__ DdSrc dd_src(const GeoJob &J, int which) {
  DdSrc S; S.data = (const uint32_t *)(which == 0 ? (const void *)J.pos : (which == 1 ? (const void *)J.uv : (const void *)J.nrm));
  S.n = S.data ? (which == 0 ? J.n_pos : (which == 1 ? J.n_uv : J.n_nrm)) : 0u; S.nw = which == 1 ? 2u : 3u; return S;
}
__device__ __forceinline__ uint64_t dd_hash(const uint32_t w[3], uint32_t nw) {
  uint64_t h = 1469598103934665603ULL;
  for (uint32_t k = 0; k < nw; k++) h = g_mix64(h ^ w[k]);
  return h;
}
__device__ __forceinline__ uint32_t dd_bin(uint64_t h, uint32_t nb) { return (uint32_t)(h >> 40) & (nb - 1); }
__device__ __forceinline__ uint32_t dd_slot(uint64_t h, uint32_t slots) { return (uint32_t)h & (slots - 1); }
// pass 1: per tile, the number of values per bin; canon[] starts as the identity
__global__ void __launch_bounds__(UVOL_BLOCK) k_dd_count(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const int which = (int)blockIdx.z; const DdSrc S = dd_src(J, which);
  const uint32_t nb = J.dd_nb[which], nblk = J.dd_nblk[which];
  if (blockIdx.x >= nblk) return;
  __shared__ uint32_t hist[DD_MAXBINS];
  for (uint32_t b = threadIdx.x; b < nb; b += UVOL_BLOCK) hist[b] = 0;
  __syncthreads();
  for (uint32_t k = 0; k < DD_TILE / UVOL_BLOCK; k++) {
    const uint32_t i = blockIdx.x * DD_TILE + k * UVOL_BLOCK + threadIdx.x;
    if (i < S.n) {
      uint32_t w[3] = { 0, 0, 0 };
      for (uint32_t q = 0; q < S.nw; q++) w[q] = S.data[(size_t)i * S.nw + q];
      atomicAdd(&hist[dd_bin(dd_hash(w, S.nw), nb)], 1u);
      J.canon[which][i] = i;
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nb; b += UVOL_BLOCK) J.dd_cnt[which][(size_t)b * nblk + blockIdx.x] = hist[b];
}
// pass 2: exclusive scan of counts[bin][tile] in bin-major order (one workgroup per frame and attribute); [nb * nblk] = n
__global__ void __launch_bounds__(UVOL_BLOCK) k_dd_scan(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const int which = (int)blockIdx.z;
  const uint32_t m = J.dd_nb[which] * J.dd_nblk[which];
  uint32_t *cnt = J.dd_cnt[which];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < m; b0 += UVOL_BLOCK) {
    const uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < m ? cnt[i] : 0, tot;
    const uint32_t ex = block_excl_scan(v, &tot);
    const uint32_t c = carry;
    if (i < m) cnt[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt[m] = carry;
}
// pass 3: scatter {index, words} into the bins (order inside a bin is arbitrary: the result is a minimum)
__global__ void __launch_bounds__(UVOL_BLOCK) k_dd_scatter(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const int which = (int)blockIdx.z; const DdSrc S = dd_src(J, which);
  const uint32_t nb = J.dd_nb[which], nblk = J.dd_nblk[which];
  if (blockIdx.x >= nblk) return;
  __shared__ uint32_t cur[DD_MAXBINS];
  for (uint32_t b = threadIdx.x; b < nb; b += UVOL_BLOCK) cur[b] = J.dd_cnt[which][(size_t)b * nblk + blockIdx.x];
  __syncthreads();
  for (uint32_t k = 0; k < DD_TILE / UVOL_BLOCK; k++) {
    const uint32_t i = blockIdx.x * DD_TILE + k * UVOL_BLOCK + threadIdx.x;
    if (i < S.n) {
      uint32_t w[3] = { 0, 0, 0 };
      for (uint32_t q = 0; q < S.nw; q++) w[q] = S.data[(size_t)i * S.nw + q];
      const uint32_t pos = atomicAdd(&cur[dd_bin(dd_hash(w, S.nw), nb)], 1u);
      J.dd_part[which][pos] = make_uint4(i, w[0], w[1], w[2]);
    }
  }
}
// pass 4: one workgroup per bin: LDS hash table slot -> (record of the first value that claimed it, lowest index of its value).
// A thread's records are fetched together (DD_PER independent 16-byte loads) and the bin's keys are staged in LDS, so a probe
// that meets an occupied slot compares against LDS: with a global read of the slot's record per probe every trip of the loop was
// two dependent round trips for the whole wave.
// Two sizes: bins of the usual load (<= ~1100 values: 100 k-vertex frames give ~780) take a 2048-slot table and 1024 staged keys = 28 KB
// of LDS; the 4096-slot / 1536-key form (50 KB) is for meshes beyond ~1 M values per attribute, whose 1024 bins hold more.  The small form
// matters beside other contexts: a workgroup that wants a third of a CU's LDS waits for it - 54 ms per 1280 frames next to the texture
// context against 9 ms per 2160 alone (profiles/r04_a_kernel_stats.csv).
template <int DD_TSLOTS, int DD_PER>
__global__ void __launch_bounds__(UVOL_BLOCK) k_dd_resolve(GeoJob *jobs, uint32_t slots) {
  constexpr uint32_t DD_KEYS = UVOL_BLOCK * DD_PER;      // keys of a bin held in LDS (the rest compares through global memory)
  JOB_OR_RETURN_UNIFORM;
  const int which = (int)blockIdx.z; const DdSrc S = dd_src(J, which);
  const uint32_t nb = J.dd_nb[which], nblk = J.dd_nblk[which];
  if (blockIdx.x >= nb || S.n == 0) return;
  const uint32_t lo = J.dd_cnt[which][(size_t)blockIdx.x * nblk], hi = J.dd_cnt[which][(size_t)(blockIdx.x + 1) * nblk];
  const uint4 *part = J.dd_part[which];
  __shared__ uint32_t t_rec[DD_TSLOTS], t_min[DD_TSLOTS];
  __shared__ uint32_t kw0[DD_KEYS], kw1[DD_KEYS], kw2[DD_KEYS];
  __shared__ uint32_t n_ins, any_dup, fail;
  for (uint32_t s = threadIdx.x; s < slots; s += UVOL_BLOCK) { t_rec[s] = 0; t_min[s] = 0xffffffffu; }
  if (threadIdx.x == 0) { n_ins = 0; any_dup = 0; fail = 0; }
#define DD_SAME(c, r) ((c) - 1 < DD_KEYS ? (kw0[(c) - 1] == (r).y && kw1[(c) - 1] == (r).z && kw2[(c) - 1] == (r).w) \
                                        : (part[lo + (c) - 1].y == (r).y && part[lo + (c) - 1].z == (r).z && part[lo + (c) - 1].w == (r).w))
  for (uint32_t e0 = lo; e0 < hi; e0 += DD_KEYS) {
    uint4 r[DD_PER];
#pragma unroll
    for (int k = 0; k < DD_PER; k++) { const uint32_t e = e0 + k * UVOL_BLOCK + threadIdx.x; r[k] = e < hi ? part[e] : make_uint4(0, 0, 0, 0); }
    if (e0 == lo) {
#pragma unroll
      for (int k = 0; k < DD_PER; k++) { const uint32_t q = k * UVOL_BLOCK + threadIdx.x; kw0[q] = r[k].y; kw1[q] = r[k].z; kw2[q] = r[k].w; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DD_PER; k++) {
      const uint32_t e = e0 + k * UVOL_BLOCK + threadIdx.x;
      if (e >= hi) continue;
      const uint32_t w[3] = { r[k].y, r[k].z, r[k].w };
      uint32_t s = dd_slot(dd_hash(w, S.nw), slots);
      for (uint32_t guard = 0;; guard++) {
        if (guard >= slots) { fail = 1; break; }
        uint32_t c = t_rec[s];
        if (c == 0) { const uint32_t old = atomicCAS(&t_rec[s], 0u, e - lo + 1); if (old == 0) { atomicMin(&t_min[s], r[k].x); atomicAdd(&n_ins, 1u); break; } c = old; }
        if (DD_SAME(c, r[k])) { atomicMin(&t_min[s], r[k].x); any_dup = 1; break; }
        s = (s + 1) & (slots - 1);
      }
    }
  }
  __syncthreads();
  if (fail || n_ins > slots - slots / 4) { if (threadIdx.x == 0) J.status = GEO_E_DD_OVERFLOW; return; }
  if (!any_dup) return;                                   // every value of the bin is unique: canon[] stays the identity
  for (uint32_t e0 = lo; e0 < hi; e0 += UVOL_BLOCK) {
    const uint32_t e = e0 + threadIdx.x;
    if (e < hi) {
      const uint4 r = part[e]; const uint32_t w[3] = { r.y, r.z, r.w };
      uint32_t s = dd_slot(dd_hash(w, S.nw), slots);
      for (uint32_t guard = 0; guard < slots; guard++) {
        const uint32_t c = t_rec[s];
        if (c == 0) break;
        if (DD_SAME(c, r)) { if (t_min[s] != r.x) J.canon[which][r.x] = t_min[s]; break; }
        s = (s + 1) & (slots - 1);
      }
    }
  }
#undef DD_SAME
}

// three ints moved as one 12-byte access
#ifdef HIPEMU
struct uvol_s3 { int32_t x, y, z; };
#else
typedef int32_t uvol_s3 __attribute__((ext_vector_type(3), aligned(4)));
#endif
// ------------------------------------------------------------------------------------------------
// Locality relabelling.  The serial walkers pay one dependent memory access per face, and what that access costs is decided by
// where the neighbouring face's record lies: in a file whose faces / vertices are stored in scan order (no relation between
// index and place on the surface) every step is an HBM miss and the gather kernels lose their coalescing - 2.3x for the whole
// path (profiles/r02_*_variant_shuffled_order).  So the frame is relabelled first: positions get new ids in Morton order of
// their coordinates (10 bits per axis over the bounding box), faces are stored in the order of their lowest new vertex id.
// Neither the ids nor the storage order reach the bitstream: vertex ids are identities, the renumbering into decoder order
// follows the walk, and the two places that DO depend on the input's face order - which unvisited face starts the next
// component, and which corner wins a non-manifold edge - keep using the original order through forig[] / s_of_o[].  The .drc
// is byte-identical with and without the relabelling (tests: shuffled and lattice storage of one surface give the same bytes).
// It is not a full sort and does not need to be: keys are binned by their top bits (count -> scan -> scatter of 8-byte records,
// LDS counters only), then one workgroup per bin orders its records by the next 11 bits with an LDS histogram; entries with
// equal prefixes stay in arbitrary order (the new ids are a performance hint, any bijection is correct).
// ------------------------------------------------------------------------------------------------
#define MS_TILE 2048
#define MS_MAXBINS 1024
#define MS_SUB 2048
__device__ __forceinline__ uint32_t ms_spread10(uint32_t x) {
  x &= 0x3ffu; x = (x | (x << 16)) & 0x030000ffu; x = (x | (x << 8)) & 0x0300f00fu; x = (x | (x << 4)) & 0x030c30c3u; x = (x | (x << 2)) & 0x09249249u; return x;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_ms_key_pos(GeoJob *jobs) {
  JOB_OR_RETURN;
  if (!J.relabel) return;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (i >= J.n_pos) return;
  uint32_t key = 0;
  for (int k = 0; k < 3; k++) {
    const float lo = g_float_unorder(J.pos_min_u[k]), hi = g_float_unorder(J.pos_max_u[k]), r = hi - lo;
    const float t = r > 0.f ? (J.pos[3 * (size_t)i + k] - lo) * (1023.0f / r) : 0.f;
    const uint32_t q = t >= 1023.f ? 1023u : (t > 0.f ? (uint32_t)t : 0u);                 // NaN -> 0
    key |= ms_spread10(q) << k;
  }
  J.ms_key[0][i] = key;
}
// Is the frame stored coherently already (consecutive faces adjacent on the surface, the vertices of a face close in index: a
// lattice, a strip-ordered export, a file that went through a vertex-cache optimiser)?  Then the relabelling would only cost its
// passes (+8 % on the lattice bench) and is skipped for this frame.  relabel: 2 = decide here, 1 = forced on, 0 = off.
__global__ void __launch_bounds__(UVOL_BLOCK) k_coherence(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  const bool on = J.status == 0;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  uint32_t share = 0, tight = 0, same = 0;
  if (on && f < J.nf_in) {
    const uint32_t a0 = J.ipos[3 * f], a1 = J.ipos[3 * f + 1], a2 = J.ipos[3 * f + 2];
    if (f > 0 && J.relabel == 2) {
      const uint32_t b0 = J.ipos[3 * f - 3], b1 = J.ipos[3 * f - 2], b2 = J.ipos[3 * f - 1];
      share = (a0 == b0 || a0 == b1 || a0 == b2 || a1 == b0 || a1 == b1 || a1 == b2 || a2 == b0 || a2 == b1 || a2 == b2) ? 1u : 0u;
      const uint32_t mx = a0 > a1 ? (a0 > a2 ? a0 : a2) : (a1 > a2 ? a1 : a2), mn = a0 < a1 ? (a0 < a2 ? a0 : a2) : (a1 < a2 ? a1 : a2);
      tight = (mx - mn) <= J.n_pos / 16u + 64u ? 1u : 0u;
    }
    // the same connectivity as the previous frame of the batch (an animated mesh of fixed topology): such frames are walked in
    // lock step, which decides how many walkers share a wave (geo_encode_batch)
    if (blockIdx.y > 0) { const GeoJob &P = jobs[blockIdx.y - 1]; if (P.nf_in == J.nf_in && P.n_pos == J.n_pos) same = (P.ipos[3 * f] == a0 && P.ipos[3 * f + 1] == a1 && P.ipos[3 * f + 2] == a2) ? 1u : 0u; }
  }
  const uint32_t s1 = block_sum(share), s2 = block_sum(tight), s3 = block_sum(same);
  if (threadIdx.x == 0 && on) { if (s1) atomicAdd(&J.coh_share, s1); if (s2) atomicAdd(&J.coh_tight, s2); if (s3) atomicAdd(&J.coh_same, s3); }
}
// per frame: relabel or not; per batch (counts[0..1]): frames that are relabelled, frames with their predecessor's connectivity
__global__ void __launch_bounds__(64) k_relabel_decide(GeoJob *jobs, int n, uint32_t *counts) {
  const int j = (int)(blockIdx.x * 64 + threadIdx.x);
  if (j >= n) return;
  GeoJob &J = jobs[j];
  if (J.relabel == 2) {
    const uint64_t nf = J.nf_in, share = J.coh_share, tight = J.coh_tight;
    J.relabel = (share * 100 >= nf * 60 && tight * 100 >= nf * 90) ? 0 : 1;
  }
  J.ms_nb[1] = ((J.n_pos ? J.n_pos - 1 : 0) >> J.ms_sh[1]) + 1; J.ms_nblk[1] = (J.nf_in + MS_TILE - 1) / MS_TILE;
  if (J.relabel) atomicAdd(&counts[0], 1u);
  if (J.coh_same == J.nf_in) atomicAdd(&counts[1], 1u);
}
__device__ __forceinline__ uint32_t ms_count_of(const GeoJob &J, int which) { return which == 0 ? J.n_pos : J.nf_in; }
__global__ void __launch_bounds__(UVOL_BLOCK) k_ms_count(GeoJob *jobs, int which) {
  JOB_OR_RETURN_UNIFORM;
  if (!J.relabel) return;
  const uint32_t nb = J.ms_nb[which], nblk = J.ms_nblk[which], sh = J.ms_sh[which], n = ms_count_of(J, which);
  if (blockIdx.x >= nblk) return;
  __shared__ uint32_t hist[MS_MAXBINS];
  for (uint32_t b = threadIdx.x; b < nb; b += UVOL_BLOCK) hist[b] = 0;
  __syncthreads();
  const uint32_t *key = J.ms_key[which];
  for (uint32_t k = 0; k < MS_TILE / UVOL_BLOCK; k++) {
    const uint32_t i = blockIdx.x * MS_TILE + k * UVOL_BLOCK + threadIdx.x;
    if (i < n) { const uint32_t kk = key[i]; if (kk != 0xffffffffu) atomicAdd(&hist[kk >> sh], 1u); }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nb; b += UVOL_BLOCK) J.ms_cnt[(size_t)b * nblk + blockIdx.x] = hist[b];
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_ms_scan(GeoJob *jobs, int which) {
  JOB_OR_RETURN_UNIFORM;
  if (!J.relabel) return;
  const uint32_t m = J.ms_nb[which] * J.ms_nblk[which];
  uint32_t *cnt = J.ms_cnt;
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < m; b0 += UVOL_BLOCK) {
    const uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < m ? cnt[i] : 0, tot;
    const uint32_t ex = block_excl_scan(v, &tot);
    const uint32_t c = carry;
    if (i < m) cnt[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt[m] = carry;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_ms_scatter(GeoJob *jobs, int which) {
  JOB_OR_RETURN_UNIFORM;
  if (!J.relabel) return;
  const uint32_t nb = J.ms_nb[which], nblk = J.ms_nblk[which], sh = J.ms_sh[which], n = ms_count_of(J, which);
  if (blockIdx.x >= nblk) return;
  __shared__ uint32_t cur[MS_MAXBINS];
  for (uint32_t b = threadIdx.x; b < nb; b += UVOL_BLOCK) cur[b] = J.ms_cnt[(size_t)b * nblk + blockIdx.x];
  __syncthreads();
  const uint32_t *key = J.ms_key[which];
  for (uint32_t k = 0; k < MS_TILE / UVOL_BLOCK; k++) {
    const uint32_t i = blockIdx.x * MS_TILE + k * UVOL_BLOCK + threadIdx.x;
    if (i < n) { const uint32_t kk = key[i]; if (kk != 0xffffffffu) { const uint32_t pos = atomicAdd(&cur[kk >> sh], 1u); J.ms_part[pos] = make_uint2(kk, i); } }
  }
}
// one workgroup per bin: order the bin's records by the next (up to) 11 key bits and hand out the final slots
__global__ void __launch_bounds__(UVOL_BLOCK) k_ms_place(GeoJob *jobs, int which) {
  JOB_OR_RETURN_UNIFORM;
  if (!J.relabel) return;
  const uint32_t nb = J.ms_nb[which], nblk = J.ms_nblk[which], sh = J.ms_sh[which];
  if (blockIdx.x >= nb) return;
  const uint32_t lo = J.ms_cnt[(size_t)blockIdx.x * nblk], hi = J.ms_cnt[(size_t)(blockIdx.x + 1) * nblk];
  const uint32_t sh2 = sh > 11u ? sh - 11u : 0u, smask = (1u << (sh - sh2)) - 1u;          // sub-key = key bits [sh2, sh)
  __shared__ uint32_t sub[MS_SUB];
  __shared__ uint32_t carry;
  for (uint32_t j = threadIdx.x; j < MS_SUB; j += UVOL_BLOCK) sub[j] = 0;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const uint2 *part = J.ms_part;
  for (uint32_t e = lo + threadIdx.x; e < hi; e += UVOL_BLOCK) atomicAdd(&sub[(part[e].x >> sh2) & smask], 1u);
  __syncthreads();
  for (uint32_t j0 = 0; j0 < MS_SUB; j0 += UVOL_BLOCK) {
    const uint32_t j = j0 + threadIdx.x;
    uint32_t v = sub[j], tot;
    const uint32_t ex = block_excl_scan(v, &tot);
    const uint32_t c = carry;
    sub[j] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  for (uint32_t e = lo + threadIdx.x; e < hi; e += UVOL_BLOCK) {
    const uint2 r = part[e];
    const uint32_t slot = lo + atomicAdd(&sub[(r.x >> sh2) & smask], 1u);
    if (which == 0) {
      J.prank[r.y] = slot;
      const float *src = J.pos + 3 * (size_t)r.y; float *dst = J.pos_s + 3 * (size_t)slot;
      dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    } else J.fperm[slot] = r.y;
  }
}
// per kept input face: its index among the kept faces in input order (the face numbering Draco's semantics refer to)
__global__ void __launch_bounds__(UVOL_BLOCK) k_face_cidx(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  if (!J.relabel) return;                                  // block-uniform
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const bool live = J.status == 0 && f < J.nf_in;
  uint32_t v = live ? J.keep[f] : 0, tot;
  const uint32_t pos = block_excl_scan(v, &tot) + (blockIdx.x <= uvol_blocks_dev(J.nf_in) ? J.bsum[blockIdx.x] : 0);
  if (live && v) J.cidx[f] = pos;
  if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) {
    const uint32_t nf = J.bsum[uvol_blocks_dev(J.nf_in)];
    J.nf = nf; J.nc = 3 * nf;
    if (nf == 0) J.status = -3;
  }
}
// stored face s <- input face fperm[s]: canonical ids (positions in their new numbering) and the maps to / from the original order
__global__ void __launch_bounds__(UVOL_BLOCK) k_relabel_faces(GeoJob *jobs) {
  JOB_OR_RETURN;
  if (!J.relabel) return;
  const uint32_t s = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (s >= J.nf) return;
  const uint32_t f = J.fperm[s];
  uvol_s3 a, b, c;
  a.x = (int32_t)J.prank[J.canon[0][J.ipos[3 * f]]]; a.y = (int32_t)J.prank[J.canon[0][J.ipos[3 * f + 1]]]; a.z = (int32_t)J.prank[J.canon[0][J.ipos[3 * f + 2]]];
  b.x = b.y = b.z = 0; c.x = c.y = c.z = 0;
  if (J.has_uv) { b.x = (int32_t)J.canon[1][J.iuv[3 * f]]; b.y = (int32_t)J.canon[1][J.iuv[3 * f + 1]]; b.z = (int32_t)J.canon[1][J.iuv[3 * f + 2]]; }
  if (J.has_nrm) { c.x = (int32_t)J.canon[2][J.inrm[3 * f]]; c.y = (int32_t)J.canon[2][J.inrm[3 * f + 1]]; c.z = (int32_t)J.canon[2][J.inrm[3 * f + 2]]; }
  *reinterpret_cast<uvol_s3 *>(J.cp + 3 * (size_t)s) = a; *reinterpret_cast<uvol_s3 *>(J.cu + 3 * (size_t)s) = b; *reinterpret_cast<uvol_s3 *>(J.cn + 3 * (size_t)s) = c;
  const uint32_t co = J.cidx[f];
  J.forig[s] = (int32_t)co; J.s_of_o[co] = (int32_t)s;
}

// per input face: canonical ids, keep flag, index validation
__global__ void __launch_bounds__(UVOL_BLOCK) k_faces(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  uint32_t keep = 0; bool bad = false;
  if (f < J.nf_in) {
    uint32_t a[3];
    for (int k = 0; k < 3; k++) {
      uint32_t ip = J.ipos[3 * f + k]; if (ip >= J.n_pos) { bad = true; ip = 0; }
      if (J.has_uv && J.iuv[3 * f + k] >= J.n_uv) bad = true;
      if (J.has_nrm && J.inrm[3 * f + k] >= J.n_nrm) bad = true;
      a[k] = J.canon[0][ip];
    }
    keep = (a[0] != a[1] && a[1] != a[2] && a[0] != a[2]) ? 1u : 0u;
    J.keep[f] = (uint8_t)keep;
    if (J.relabel) {                                                     // sort key of the face: its lowest NEW vertex id (dropped faces are left out)
      uint32_t k0 = 0xffffffffu;
      if (keep && !bad) { const uint32_t r0 = J.prank[a[0]], r1 = J.prank[a[1]], r2 = J.prank[a[2]]; k0 = r0 < r1 ? r0 : r1; k0 = r2 < k0 ? r2 : k0; }
      J.ms_key[1][f] = k0;
    }
  }
  const uint32_t tot = block_sum(keep);                                  // block sums of the keep flags (was a k_scan_blocks pass)
  if (threadIdx.x == 0 && blockIdx.x < uvol_blocks_dev(J.nf_in)) J.bsum[blockIdx.x] = tot;
  if (bad) J.status = -2;                                                // after the barriers: a wave that has not started yet leaves at once when it sees it
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_compact_faces(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  if (J.relabel) return;                                   // block-uniform: k_face_cidx + k_relabel_faces store the faces instead
  uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  bool live = J.status == 0 && f < J.nf_in;
  uint32_t v = live ? J.keep[f] : 0, tot;
  uint32_t pos = block_excl_scan(v, &tot) + (blockIdx.x <= uvol_blocks_dev(J.nf_in) ? J.bsum[blockIdx.x] : 0);
  if (live && v) {                                        // one 12-byte store per array and face instead of three dword stores
    uvol_s3 a, b, c;
    a.x = (int32_t)J.canon[0][J.ipos[3 * f]]; a.y = (int32_t)J.canon[0][J.ipos[3 * f + 1]]; a.z = (int32_t)J.canon[0][J.ipos[3 * f + 2]];
    b.x = b.y = b.z = 0; c.x = c.y = c.z = 0;
    if (J.has_uv) { b.x = (int32_t)J.canon[1][J.iuv[3 * f]]; b.y = (int32_t)J.canon[1][J.iuv[3 * f + 1]]; b.z = (int32_t)J.canon[1][J.iuv[3 * f + 2]]; }
    if (J.has_nrm) { c.x = (int32_t)J.canon[2][J.inrm[3 * f]]; c.y = (int32_t)J.canon[2][J.inrm[3 * f + 1]]; c.z = (int32_t)J.canon[2][J.inrm[3 * f + 2]]; }
    *reinterpret_cast<uvol_s3 *>(J.cp + 3 * (size_t)pos) = a; *reinterpret_cast<uvol_s3 *>(J.cu + 3 * (size_t)pos) = b; *reinterpret_cast<uvol_s3 *>(J.cn + 3 * (size_t)pos) = c;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) {
    uint32_t nf = J.bsum[uvol_blocks_dev(J.nf_in)];
    J.nf = nf; J.nc = 3 * nf;
    if (nf == 0) J.status = -3;
  }
}

// ------------------------------------------------------------------------------------------------
// K3: opposite corners.  opp[c] = the lowest corner facing the reversed edge, if c itself is the lowest corner on its own
// directed edge (a -> b) = (vertex of next(c), vertex of prev(c)); otherwise none.  Directed edges are bucketed by their
// from-vertex (count -> scan -> fill), so a corner reads two short contiguous buckets (its own edge's and the reversed
// edge's, ~valence entries each) out of a 4.8 MB array with the mesh's own locality - a 24 MB open-addressing hash table
// of 64-bit keys did the same with 242 MB of scattered HBM traffic per frame, 29 % of the whole frame's (r01_i PMC passes).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(UVOL_BLOCK) k_he_count(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t c0 = blockIdx.x * (UVOL_BLOCK * GEO_ILP) + threadIdx.x, nc = J.nc;
  uint32_t a[GEO_ILP];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK; a[k] = c < nc ? (uint32_t)J.cp[g_nxt(c)] : 0xffffffffu; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) if (a[k] != 0xffffffffu) atomicAdd(&J.he_start[a[k]], 1u);
}
// one workgroup per frame: exclusive scan of the per-vertex counts in place, cursor = start
__global__ void __launch_bounds__(UVOL_BLOCK) k_he_scan(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  if (J.status != 0) return;
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const uint32_t n = J.n_pos;
  for (uint32_t b0 = 0; b0 < n; b0 += UVOL_BLOCK) {
    const uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < n ? J.he_start[i] : 0, tot;
    const uint32_t ex = block_excl_scan(v, &tot);
    const uint32_t c = carry;
    if (i < n) { J.he_start[i] = c + ex; J.he_cur[i] = c + ex; }
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_he_fill(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t c0 = blockIdx.x * (UVOL_BLOCK * GEO_ILP) + threadIdx.x, nc = J.nc;
  uint32_t a[GEO_ILP], b[GEO_ILP], slot[GEO_ILP];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK; const bool in = c < nc; a[k] = in ? (uint32_t)J.cp[g_nxt(c)] : 0xffffffffu; b[k] = in ? (uint32_t)J.cp[g_prv(c)] : 0u; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) slot[k] = a[k] != 0xffffffffu ? atomicAdd(&J.he_cur[a[k]], 1u) : 0u;
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) if (a[k] != 0xffffffffu) J.he_ent[slot[k]] = ((unsigned long long)b[k] << 32) | (unsigned long long)(c0 + k * UVOL_BLOCK);
}
// Partitioned form of the bucket build (the default): the count / fill kernels above post two device-scope atomics per corner
// (1.2 M per 200 k-face frame, memory-side) and fill the buckets with scattered 8-byte stores (17 MB of write traffic for a
// 4.8 MB array).  Here the half-edges are partitioned by ranges of `he_vpb` from-vertices (count -> scan -> scatter of 12-byte
// {from, to, corner} records), then ONE workgroup per range counts, scans and fills its buckets in LDS and writes he_start /
// he_cur / he_ent for its range contiguously.  Bucket contents are the same sets as before; their order is arbitrary either way.
#define HE_TILE 2048                        // corners per workgroup in the count / scatter passes
#define HE_MAXBINS 1024
#define HE_MAXVPB 4096                      // from-vertices per range (LDS counters)
__global__ void __launch_bounds__(UVOL_BLOCK) k_hp_count(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const uint32_t nb = J.he_nb, nblk = J.he_nblk, nc = J.nc; uint32_t sh = 9; while ((1u << sh) < J.he_vpb) sh++;
  if (blockIdx.x >= nblk) return;
  __shared__ uint32_t hist[HE_MAXBINS];
  for (uint32_t b = threadIdx.x; b < nb; b += UVOL_BLOCK) hist[b] = 0;
  __syncthreads();
  uint32_t a[HE_TILE / UVOL_BLOCK];
#pragma unroll
  for (int k = 0; k < HE_TILE / UVOL_BLOCK; k++) { const uint32_t c = blockIdx.x * HE_TILE + k * UVOL_BLOCK + threadIdx.x; a[k] = c < nc ? (uint32_t)J.cp[g_nxt(c)] : 0xffffffffu; }
#pragma unroll
  for (int k = 0; k < HE_TILE / UVOL_BLOCK; k++) if (a[k] != 0xffffffffu) atomicAdd(&hist[a[k] >> sh], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nb; b += UVOL_BLOCK) J.he_cnt[(size_t)b * nblk + blockIdx.x] = hist[b];
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_hp_scan(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const uint32_t m = J.he_nb * J.he_nblk;
  uint32_t *cnt = J.he_cnt;
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < m; b0 += UVOL_BLOCK) {
    const uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < m ? cnt[i] : 0, tot;
    const uint32_t ex = block_excl_scan(v, &tot);
    const uint32_t c = carry;
    if (i < m) cnt[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt[m] = carry;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_hp_scatter(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const uint32_t nb = J.he_nb, nblk = J.he_nblk, nc = J.nc; uint32_t sh = 9; while ((1u << sh) < J.he_vpb) sh++;
  if (blockIdx.x >= nblk) return;
  __shared__ uint32_t cur[HE_MAXBINS];
  for (uint32_t b = threadIdx.x; b < nb; b += UVOL_BLOCK) cur[b] = J.he_cnt[(size_t)b * nblk + blockIdx.x];
  __syncthreads();
  uint32_t a[HE_TILE / UVOL_BLOCK], bb[HE_TILE / UVOL_BLOCK];
#pragma unroll
  for (int k = 0; k < HE_TILE / UVOL_BLOCK; k++) {
    const uint32_t c = blockIdx.x * HE_TILE + k * UVOL_BLOCK + threadIdx.x; const bool in = c < nc;
    a[k] = in ? (uint32_t)J.cp[g_nxt(c)] : 0xffffffffu; bb[k] = in ? (uint32_t)J.cp[g_prv(c)] : 0u;
  }
#pragma unroll
  for (int k = 0; k < HE_TILE / UVOL_BLOCK; k++) {
    if (a[k] == 0xffffffffu) continue;
    const uint32_t pos = atomicAdd(&cur[a[k] >> sh], 1u);
    uvol_s3 r; r.x = (int32_t)a[k]; r.y = (int32_t)bb[k]; r.z = (int32_t)(blockIdx.x * HE_TILE + k * UVOL_BLOCK + threadIdx.x);
    *reinterpret_cast<uvol_s3 *>(J.he_part + 3 * (size_t)pos) = r;
  }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_hp_build(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const uint32_t nb = J.he_nb, nblk = J.he_nblk, vpb = J.he_vpb;
  if (blockIdx.x >= nb) return;
  const uint32_t lo = J.he_cnt[(size_t)blockIdx.x * nblk], hi = J.he_cnt[(size_t)(blockIdx.x + 1) * nblk];
  const uint32_t v0 = blockIdx.x * vpb, nv = v0 < J.n_pos ? (J.n_pos - v0 < vpb ? J.n_pos - v0 : vpb) : 0u;
  __shared__ uint32_t cv[HE_MAXVPB];
  __shared__ uint32_t carry;
  for (uint32_t j = threadIdx.x; j < vpb; j += UVOL_BLOCK) cv[j] = 0;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t e = lo + threadIdx.x; e < hi; e += UVOL_BLOCK) atomicAdd(&cv[J.he_part[3 * (size_t)e] - v0], 1u);
  __syncthreads();
  for (uint32_t j0 = 0; j0 < vpb; j0 += UVOL_BLOCK) {                    // exclusive scan in place; bucket bounds for the range
    const uint32_t j = j0 + threadIdx.x;
    uint32_t v = cv[j], tot;
    const uint32_t ex = block_excl_scan(v, &tot);
    const uint32_t c = carry;
    cv[j] = c + ex;
    if (j < nv) { J.he_start[v0 + j] = lo + c + ex; J.he_cur[v0 + j] = lo + c + ex + v; }
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  for (uint32_t e = lo + threadIdx.x; e < hi; e += UVOL_BLOCK) {
    const uvol_s3 r = *reinterpret_cast<const uvol_s3 *>(J.he_part + 3 * (size_t)e);
    const uint32_t slot = lo + atomicAdd(&cv[(uint32_t)r.x - v0], 1u);
    J.he_ent[slot] = ((unsigned long long)(uint32_t)r.y << 32) | (unsigned long long)(uint32_t)r.z;
  }
}
// lowest corner on the directed edge (from -> to), or -1; the order inside a bucket is arbitrary, the minimum is not
__device__ __forceinline__ int he_find(const GeoJob &J, uint32_t from, uint32_t to) {
  const uint32_t s = J.he_start[from], e = J.he_cur[from];
  uint32_t best = 0xffffffffu;
  for (uint32_t i = s; i < e; i++) { const unsigned long long v = J.he_ent[i]; if ((uint32_t)(v >> 32) == to) { const uint32_t cc = (uint32_t)v; best = cc < best ? cc : best; } }
  return best == 0xffffffffu ? -1 : (int)best;
}
// bucket bounds of both directed edges of GEO_ILP corners are fetched before any bucket is scanned
__global__ void __launch_bounds__(UVOL_BLOCK) k_edge_match(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t c0 = blockIdx.x * (UVOL_BLOCK * GEO_ILP) + threadIdx.x, nc = J.nc;
  uint32_t a[GEO_ILP], b[GEO_ILP], sa[GEO_ILP], ea[GEO_ILP], sb[GEO_ILP], eb[GEO_ILP];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK, cc = c < nc ? c : 0u; a[k] = (uint32_t)J.cp[g_nxt(cc)]; b[k] = (uint32_t)J.cp[g_prv(cc)]; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { sa[k] = J.he_start[a[k]]; ea[k] = J.he_cur[a[k]]; sb[k] = J.he_start[b[k]]; eb[k] = J.he_cur[b[k]]; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) {
    const uint32_t c = c0 + k * UVOL_BLOCK;
    if (c >= nc) continue;
    uint32_t self = 0xffffffffu, o = 0xffffffffu;      // (fetching the first eight entries of both buckets at once was slower: 24 vs 20 ms)
    // "lowest corner" means lowest in the ORIGINAL face order: only looked up when an edge has several corners (non-manifold)
    const bool rl = J.relabel != 0;
#define EM_LOWER(x, y) (rl ? (3u * (uint32_t)J.forig[(x) / 3u] + (x) % 3u < 3u * (uint32_t)J.forig[(y) / 3u] + (y) % 3u) : ((x) < (y)))
    for (uint32_t i = sa[k]; i < ea[k]; i++) { const unsigned long long v = J.he_ent[i]; if ((uint32_t)(v >> 32) == b[k]) { const uint32_t cc = (uint32_t)v; if (self == 0xffffffffu || EM_LOWER(cc, self)) self = cc; } }
    for (uint32_t i = sb[k]; i < eb[k]; i++) { const unsigned long long v = J.he_ent[i]; if ((uint32_t)(v >> 32) == a[k]) { const uint32_t cc = (uint32_t)v; if (o == 0xffffffffu || EM_LOWER(cc, o)) o = cc; } }
#undef EM_LOWER
    J.opp[c] = (self == c && o != 0xffffffffu) ? (int)o : GEO_INV;
  }
}

// ------------------------------------------------------------------------------------------------
// Vertices.  A corner-table vertex is a fan of corners around a position.  On a manifold mesh that IS the position, so the
// vertex id of a corner is its canonical position id (cp[]); only a position shared by several fans (non-manifold vertex)
// needs more ids.  One thread per POSITION walks one fan of its corner bucket (the half-edge buckets of K3 list every corner
// at the position): if the fan has as many corners as the bucket, the position is one vertex — open flag, ring size and the
// corners' ids follow without walking from every corner (k_fans did that: valence x more dependent loads, 20 % of the
// geometry time at 2160 frames per launch).  Otherwise every fan is walked from its representative corner and all but the
// first get ids n_pos + k.  Ids are identities, not an order: nothing in the bitstream depends on how vertices are numbered
// (visited bitmaps, valences and entry maps are keyed by them), so ids may have holes (unused positions) and the extra ids of
// non-manifold fans may be handed out in any order.
// Table 1 (decoder-order base table) re-uses these ids through the corner renumbering; the attribute tables split only the
// vertices an interior seam touches (k_aseg_a / k_aseg_b), every other vertex keeps its base id.
// ------------------------------------------------------------------------------------------------
// fan of corner c in table T: representative (left-most corner of an open fan, lowest corner of a closed one), size, open flag
__device__ inline int fan_probe(const GTab &T, int c, int limit, int &cnt, bool &open) {
  int l = c, mn = c; cnt = 1; open = true;
  for (;;) { const int nl = gt_swl(T, l); if (nl < 0) break; if (nl == c) { open = false; break; } l = nl; mn = l < mn ? l : mn; if (++cnt > limit) return -1; }
  if (!open) return mn;
  for (int a = gt_swr(T, c); a >= 0; a = gt_swr(T, a)) if (++cnt > limit) return -1;
  return l;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_vert0(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t p = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (p >= J.n_pos) return;
  const uint32_t s = J.he_start[p], n = J.he_cur[p] - s;
  if (n == 0) { J.ring_d[p] = 0; J.vopen_d[0][p] = 0; return; }        // position no face uses: its id stays a hole
  GTab T; T.opp = J.opp; T.seam = nullptr;
  const int c0 = g_nxt((int)(uint32_t)J.he_ent[s]);                    // bucket entry = corner facing the edge; its next corner sits at p
  int cnt; bool open;
  if (fan_probe(T, c0, (int)n, cnt, open) < 0) { J.status = -22; return; }
  if ((uint32_t)cnt == n) {                                             // one fan: the position is the vertex
    for (uint32_t i = 0; i < n; i++) J.vert[g_nxt((int)(uint32_t)J.he_ent[s + i])] = (int32_t)p;
    J.vopen_d[0][p] = open ? 1 : 0; J.ring_d[p] = (int32_t)(open ? n + 1 : n);
    atomicAdd(&J.nverts, 1u);
    return;
  }
  bool first = true;                                                    // non-manifold vertex: one id per fan
  for (uint32_t i = 0; i < n; i++) {
    const int c = g_nxt((int)(uint32_t)J.he_ent[s + i]);
    const int rep = fan_probe(T, c, (int)n, cnt, open);
    if (rep < 0) { J.status = -22; return; }
    if (rep != c) continue;                                             // each fan is handled once, from its representative
    const uint32_t id = first ? p : J.n_pos + atomicAdd(&J.extra_v, 1u);
    first = false;
    atomicAdd(&J.nverts, 1u);
    int a = rep;
    for (int k = 0; k < cnt; k++) { J.vert[a] = (int32_t)id; a = open ? gt_swr(T, a) : gt_swl(T, a); }
    if (id < J.ecap) { J.vopen_d[0][id] = open ? 1 : 0; J.ring_d[id] = open ? cnt + 1 : cnt; }
  }
  if (first) J.status = -22;
}

// ------------------------------------------------------------------------------------------------
// K4: valence edgebreaker — split into
//   k_pack0        (parallel)  per-corner records {vertex<<1|open, right, left[, opposite]} (8 or 16 bytes, RecOps) indexed by corner code
//                              4*face+k, so a walker step is ONE load and no division / select
//   k_eb_walk      (serial)    MeshEdgebreakerEncoderImpl::EncodeConnectivity traversal only: symbols + processed corners;
//                              visited faces / vertices are bitmaps in LDS (k_face_time inverts proc[] afterwards)
//   k_eb_events    (parallel)  topology-split events from (symbol, neighbour symbol) pairs, order-preserving compaction
//   k_eb_valence   (1 wave)    MeshEdgebreakerTraversalValenceEncoder bookkeeping replayed over the known symbol
//                              sequence: runs between split symbols are resolved by all 64 lanes at once
//   k_eb_ctx       (1 wave)    ballot-ordered scatter of the symbols into the 6 valence-context streams
// (SURVEY A.3 / A.10).  The serial kernels run one frame per workgroup; a batch keeps that many CUs busy.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dense_table_live(const GeoJob &J, int which) { const int ai = which >= 2 ? which - 2 : 0; return !(which >= 2 && (ai >= J.nad || !J.interior_seams[ai])); }
// the three records of face f (r[k] = opposite corner of corner k, vc[k] = vertex << 1 | open) in either format
// r8: 0 = 16 bytes per corner, 1 = 8 bytes per corner, 2 = ONE 16-byte record per FACE (f16_*, the lane-per-walker kernels)
__device__ __forceinline__ void pack_face_records(int32_t *rec, uint32_t f, const int vc[3], const int r[3], int r8) {
  if (r8 == 2) {
    // {vertex << 1 | open} x 3 in the low 64 bits (bit 63: the walker's face-visited flag), the opposite corner codes x 3 in the high
    // 64 bits, 21-bit fields: what the three 8-byte corner records hold (each opposite twice) in half the bytes, and the flag in it
    uint64_t lo = 0, hi = 0;
    for (int k = 0; k < 3; k++) { lo |= (uint64_t)((uint32_t)vc[k] & 0x1fffffu) << (21 * k); hi |= (uint64_t)((uint32_t)code_of_corner(r[k]) & 0x1fffffu) << (21 * k); }
    reinterpret_cast<uint4 *>(rec)[f] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
  } else if (r8) {
    uint4 *dst = reinterpret_cast<uint4 *>(rec) + 2 * (size_t)f;      // the face's 32-byte block as two 16-byte stores
    uint2 q[3];
    for (int k = 0; k < 3; k++) {
      const uint32_t R = (uint32_t)code_of_corner(r[(k + 1) % 3]) & 0x1fffffu, L = (uint32_t)code_of_corner(r[(k + 2) % 3]) & 0x1fffffu;
      q[k] = make_uint2(((uint32_t)vc[k] & 0x1fffffu) | (R << 21), (R >> 11) | (L << 10));
    }
    dst[0] = make_uint4(q[0].x, q[0].y, q[1].x, q[1].y);
    dst[1] = make_uint4(q[2].x, q[2].y, 0u, 0u);    // 4th slot of the face's block: "face visited" flag of the lane-per-walker kernels
  } else {
    int4 *dst = reinterpret_cast<int4 *>(rec) + 4 * (size_t)f;
    for (int k = 0; k < 3; k++) dst[k] = make_int4(vc[k], code_of_corner(r[(k + 1) % 3]), code_of_corner(r[(k + 2) % 3]), code_of_corner(r[k]));
    dst[3] = make_int4(0, 0, 0, 0);
  }
}
// decode path (geom_decode.hip prepares vert / vopen_d itself): which: 1 new base, 2/3 attribute tables (DFS)
__global__ void __launch_bounds__(UVOL_BLOCK) k_pack_faces(GeoJob *jobs, int which, int r8) {
  JOB_OR_RETURN;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f >= J.nf) return;
  const int ai = which >= 2 ? which - 2 : 0;
  if (which >= 2 && (ai >= J.nad || !J.interior_seams[ai])) return;
  const int32_t *opp = which == 0 ? J.opp : J.nopp;
  const uint8_t *seam = which >= 2 ? J.seam[ai] : nullptr;
  const int32_t *vert = which == 0 ? J.vert : (which == 1 ? J.bvert : J.avert[ai]);
  const uint8_t *vopen = J.vopen_d[which];
  int r[3], vc[3];
  for (int k = 0; k < 3; k++) {
    const int c = 3 * (int)f + k;
    r[k] = (seam && seam[c]) ? GEO_INV : opp[c];
    const int v = vert[c];
    vc[k] = (v << 1) | (vopen[v] ? 1 : 0);
  }
  pack_face_records(J.rec[which], f, vc, r, r8);
  if (which == 0) J.face_time[f] = -1;            // faces that start a component without a symbol keep -1 (see k_face_time)
}
// encoder, table 0 (old order): records for the edgebreaker walk; also publishes the size of the vertex id space
__global__ void __launch_bounds__(UVOL_BLOCK) k_pack0(GeoJob *jobs, int r8) {
  JOB_OR_RETURN;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f == 0) { const uint32_t tot = J.n_pos + J.extra_v; J.nverts_t[0] = tot; J.nverts_t[1] = tot; if (tot > J.ecap) J.status = GEO_E_WS_OVERFLOW; }
  if (f >= J.nf) return;
  int r[3], vc[3];
  for (int k = 0; k < 3; k++) {
    const int c = 3 * (int)f + k;
    r[k] = J.opp[c];
    const uint32_t v = (uint32_t)J.vert[c];
    vc[k] = (int)((v << 1) | ((v < J.ecap && J.vopen_d[0][v]) ? 1u : 0u));
  }
  pack_face_records(J.rec[0], f, vc, r, r8);
  J.face_time[f] = -1;                            // faces that start a component without a symbol keep -1 (see k_face_time)
}
// encoder, tables 1..3 (decoder order; table = 1 + blockIdx.z): base table and the attribute tables that have interior seams.
// An attribute vertex an interior seam does not touch keeps its base id and open flag; the segments of the others are open.
__global__ void __launch_bounds__(UVOL_BLOCK) k_pack3(GeoJob *jobs, int r8) {
  JOB_OR_RETURN;
  const int which = 1 + (int)blockIdx.z;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f >= J.nf || !dense_table_live(J, which)) return;
  const int ai = which >= 2 ? which - 2 : 0;
  const uint8_t *seam = which >= 2 ? J.seam[ai] : nullptr;
  const int32_t *vert = which == 1 ? J.bvert : J.avert[ai];
  const uint32_t nbase = J.nverts_t[0];
  int r[3], vc[3];
  for (int k = 0; k < 3; k++) {
    const int c = 3 * (int)f + k;
    r[k] = (seam && seam[c]) ? GEO_INV : J.nopp[c];
    const uint32_t v = (uint32_t)vert[c];
    vc[k] = (int)((v << 1) | ((v >= nbase || J.vopen_d[0][v]) ? 1u : 0u));
  }
  pack_face_records(J.rec[which], f, vc, r, r8);
}

// typed-pointer helpers for the one-lane walkers (P = UVOL_G / UVOL_L pointer)
#ifdef HIPEMU
typedef int4 uvol_i4;
#else
typedef int uvol_i4 __attribute__((ext_vector_type(4)));      // loadable through an address-space-qualified pointer
#endif
// the three live words {vertex, right, left} of a corner record as ONE 12-byte load: a prefetched 16-byte load would leave
// its dead 4th register free for the allocator to reuse at once, which forces a wait right behind the load
#ifdef HIPEMU
struct uvol_i3 { int x, y, z; };
template <typename P> __device__ __forceinline__ uvol_i3 rec3(P rec, int code) { const uvol_i4 q = rec[code]; uvol_i3 r; r.x = q.x; r.y = q.y; r.z = q.z; return r; }
#else
typedef int uvol_i3 __attribute__((ext_vector_type(3)));
template <typename P> __device__ __forceinline__ uvol_i3 rec3(P rec, int code) { return *(UVOL_G(const uvol_i3))(rec + code); }
#endif
// bitmap words: LDS pointers read with ds_read; global pointers with a workgroup-scope atomic load (sc0: not served from a
// possibly stale per-CU L1 line - the bits are set with atomic ORs performed in L2 - but, unlike the device-scope load used
// before, served by this XCD's L2 instead of the memory-side cache: one walker wave is the only reader and writer of its bitmap)
__device__ __forceinline__ uint32_t pword(UVOL_L(uint32_t) w, int k) { return w[k]; }
#ifndef HIPEMU
__device__ __forceinline__ uint32_t pword(UVOL_G(uint32_t) w, int k) { return __hip_atomic_load(&w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
template <typename P> __device__ __forceinline__ bool pbit_get(P w, int i) { return (pword(w, i >> 5) >> (i & 31)) & 1u; }
template <typename P> __device__ __forceinline__ void pbit_set(P w, int i) { UVOL_OR_NORET(&w[i >> 5], 1u << (i & 31)); }   // fire and forget
__device__ __forceinline__ int code_nxt(int x) { return (x & 3) == 2 ? x - 2 : x + 1; }
__device__ __forceinline__ int code_prv(int x) { return (x & 3) == 0 ? x + 2 : x - 1; }
__device__ __forceinline__ int corner_of_code(int x) { return 3 * (x >> 2) + (x & 3); }

// Corner records in two formats.  R8 = false: 16 bytes {vertex<<1|open, right, left, opposite} (any mesh size).  R8 = true:
// 8 bytes, three 21-bit fields {vertex<<1|open : 0..20, right : 21..41, left : 42..62} (codes and ids < 2^20, -1 = all
// ones), used whenever the batch allows it: a 128-byte line then holds the records of four faces instead of two, so more of
// a walker's dependent loads hit a line a neighbouring face already brought in, and the walkers and k_pack move half the
// bytes.  The opposite corner is not stored: opposite(k) = right field of the record of corner (k + 2) % 3.
#ifdef HIPEMU
struct uvol_u2 { uint32_t x, y; };
#else
typedef uint32_t uvol_u2 __attribute__((ext_vector_type(2)));
#endif
__device__ __forceinline__ void rec8_dec(uint32_t lo, uint32_t hi, int &vi, int &rc, int &lc) {
  vi = (int)(lo & 0x1fffffu);
  rc = (int)(((lo >> 21) | (hi << 11)) << 11) >> 11;
  lc = (int)(hi << 1) >> 11;
}
template <bool R8> struct RecOps;
template <> struct RecOps<false> {
  typedef UVOL_G(const uvol_i4) Ptr; typedef uvol_i3 Pre;
  static __device__ __forceinline__ Ptr ptr(const int32_t *p) { return UVOL_TO_G(const uvol_i4, reinterpret_cast<const uvol_i4 *>(p)); }
  static __device__ __forceinline__ void get(Ptr rec, int code, int &vi, int &rc, int &lc) { const uvol_i4 q = rec[code]; vi = q.x; rc = q.y; lc = q.z; }
  static __device__ __forceinline__ Pre pre(Ptr rec, int code) { return rec3(rec, code); }
  static __device__ __forceinline__ void take(const Pre &p, int &vi, int &rc, int &lc) { vi = UVOL_READFIRST(p.x); const int l_ = UVOL_READFIRST(p.z); rc = UVOL_READFIRST(p.y); lc = l_; }
};
template <> struct RecOps<true> {
  typedef UVOL_G(const uvol_u2) Ptr; typedef uvol_u2 Pre;
  static __device__ __forceinline__ Ptr ptr(const int32_t *p) { return UVOL_TO_G(const uvol_u2, reinterpret_cast<const uvol_u2 *>(p)); }
  static __device__ __forceinline__ void get(Ptr rec, int code, int &vi, int &rc, int &lc) { const uvol_u2 q = rec[code]; rec8_dec(q.x, q.y, vi, rc, lc); }
  static __device__ __forceinline__ Pre pre(Ptr rec, int code) { return rec[code]; }
  static __device__ __forceinline__ void take(const Pre &p, int &vi, int &rc, int &lc) { const uint32_t lo = (uint32_t)UVOL_READFIRST(p.x), hi = (uint32_t)UVOL_READFIRST(p.y); rec8_dec(lo, hi, vi, rc, lc); }
};

// Output staging of the one-lane LDS walkers.  gfx950 has ONE counter (vmcnt) for loads and stores and the compiler treats a queue
// that holds both as unordered: with a store outstanding, the wait for the record the next step needs becomes vmcnt(0) and also
// waits for the acknowledgement of the proc[] / symb[] (order[]) stores issued a moment ago - a second memory round trip per face
// on top of the record load.  The walkers therefore write their output streams to LDS (lgkmcnt) and flush WALK_STG entries at a
// time with 16-byte stores: one store acknowledgement per WALK_STG faces instead of one per face.
#define WALK_STG 256                                    // staged entries (multiple of 16)
#define WALK_STG_DWORDS (WALK_STG + WALK_STG / 4)       // int32 entries + one byte per entry
struct WalkStage {
  UVOL_L(int32_t) w; UVOL_L(uint8_t) b;
  __device__ __forceinline__ void init(UVOL_L(uint32_t) lds) { w = (UVOL_L(int32_t))lds; b = (UVOL_L(uint8_t))(lds + WALK_STG); }
  // entries [n - WALK_STG, n) of the streams leave when n reaches a multiple of WALK_STG (16-byte aligned: arrays are 256-byte aligned)
  __device__ __forceinline__ void flush_words(UVOL_G(int32_t) dst, int n) {
    UVOL_G(uvol_i4) d = (UVOL_G(uvol_i4))(dst + (n - WALK_STG)); UVOL_L(const uvol_i4) s = (UVOL_L(const uvol_i4))w;
#pragma unroll 8
    for (int i = 0; i < WALK_STG / 4; i++) d[i] = s[i];
  }
  __device__ __forceinline__ void flush_bytes(UVOL_G(uint8_t) dst, int n) {
    UVOL_G(uvol_i4) d = (UVOL_G(uvol_i4))(dst + (n - WALK_STG)); UVOL_L(const uvol_i4) s = (UVOL_L(const uvol_i4))b;
#pragma unroll 8
    for (int i = 0; i < WALK_STG / 16; i++) d[i] = s[i];
  }
  __device__ __forceinline__ void tail_words(UVOL_G(int32_t) dst, int n) { for (int i = n & ~(WALK_STG - 1); i < n; i++) dst[i] = w[i & (WALK_STG - 1)]; }
  __device__ __forceinline__ void tail_bytes(UVOL_G(uint8_t) dst, int n) { for (int i = n & ~(WALK_STG - 1); i < n; i++) dst[i] = b[i & (WALK_STG - 1)]; }
};
// Edgebreaker walk, one lane per frame: typed pointers (global_* / ds_* instructions, exactly counted waits), no scatter
// stores (face_time is rebuilt from proc[] by k_face_time).  Per face: ONE 8- or 16-byte record read from HBM — the dependent
// access that bounds the walk —, one sequential proc/symb store pair, a fire-and-forget ds_or for the face bit and one
// LDS round trip for the vertex / neighbour bits.  Corners are carried as codes (4 * face + k).
template <bool R8, typename FB, typename VB>
__device__ __forceinline__ void eb_walk_lane0(GeoJob &J, FB fbits, VB vbits, UVOL_L(uint32_t) stg_lds) {
  typedef RecOps<R8> RO;
  const int nf = (int)J.nf;
  const typename RO::Ptr rec = RO::ptr(J.rec[0]);
  UVOL_G(int32_t) proc = UVOL_TO_G(int32_t, J.proc); UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.stack);
  UVOL_G(int32_t) initc = UVOL_TO_G(int32_t, J.initc);
  UVOL_G(uint8_t) symb = UVOL_TO_G(uint8_t, J.symb); UVOL_G(uint8_t) start_bits = UVOL_TO_G(uint8_t, J.start_bits);
  const int dz = UVOL_LANE_ZERO();
  WalkStage stg; stg.init(stg_lds);
  int nproc = 0, ninit = 0, nstart = 0, nsplit = 0;
#define W_EMIT(SYM) do { stg.b[nproc & (WALK_STG - 1)] = (uint8_t)(SYM); nproc++; if ((nproc & (WALK_STG - 1)) == 0) { stg.flush_words(proc, nproc); stg.flush_bytes(symb, nproc); } } while (0)
  enum { T_C = 0, T_S = 1, T_L = 3, T_R = 5, T_E = 7 };
  const bool rl = J.relabel != 0; UVOL_G(const int32_t) s_of_o = UVOL_TO_G(const int32_t, J.s_of_o);
  for (int fo = 0; fo < nf; fo++) {
    // component starts, in the ORIGINAL face order (a relabelled frame maps it to the stored face); fully visited words of the
    // face bitmap are skipped 32 faces at a time where stored order = original order
    int f0 = fo;
    if (rl) { if (nproc + ninit >= nf) break; f0 = s_of_o[fo]; }
    else if ((fo & 31) == 0) { while (fo + 32 <= nf && pword(fbits, fo >> 5) == 0xffffffffu) fo += 32; if (fo >= nf) break; f0 = fo; }
    if (pbit_get(fbits, f0)) continue;
    int v0[3], r0_[3], l0_[3];
    for (int k = 0; k < 3; k++) RO::get(rec, 4 * f0 + k, v0[k], r0_[k], l0_[k]);
    const int o0[3] = { r0_[2], r0_[0], r0_[1] };                       // opposite(k) = right field of corner (k + 2) % 3
    int interior = 1, start = 4 * f0;
    for (int k = 0; k < 3; k++) {
      if (o0[k] < 0) { interior = 0; start = 4 * f0 + k; break; }
      if (v0[k] & 1) {                // boundary vertex: swing right to the boundary edge
        int ci = 4 * f0 + k, rc = ci;
        while (rc >= 0) { ci = rc; int v_, r_, o; RO::get(rec, rc, v_, r_, o); rc = o < 0 ? -1 : code_prv(o); }
        interior = 0; start = code_prv(ci); break;
      }
    }
    start_bits[nstart] = (uint8_t)interior;
    nstart++;
    int from;
    if (interior) {
      pbit_set(vbits, v0[0] >> 1); pbit_set(vbits, v0[1] >> 1); pbit_set(vbits, v0[2] >> 1);
      pbit_set(fbits, f0);
      initc[ninit] = 3 * f0 + 1;
      ninit++;
      from = o0[1];
      if (from < 0 || pbit_get(fbits, from >> 2)) continue;
    } else from = start;
    int sp = 0;
    stack[sp] = from;
    sp++;
    int top = from;                                   // value at stack[sp-1] when known without a load
    bool top_known = true;
    while (sp > 0) {
      int x = top_known ? top : stack[sp - 1];
      top_known = false;
      if (x < 0 || pbit_get(fbits, x >> 2)) { sp--; continue; }
      int vi, rcn, lcn;
      RO::get(rec, x, vi, rcn, lcn);
      for (;;) {
        const int face = x >> 2;
        // both records this step can move to are requested now and taken (readfirstlane) only by the branch that goes there
        const typename RO::Pre pR = RO::pre(rec, (rcn < 0 ? x : rcn) + dz), pL = RO::pre(rec, (lcn < 0 ? x : lcn) + dz);
        stg.w[nproc & (WALK_STG - 1)] = 3 * face + (x & 3);
        pbit_set(fbits, face);
        const int v = vi >> 1;
        // the three bitmap words this step can need, read together (one LDS round trip)
        const uint32_t vw_ = pword(vbits, v >> 5);
        const uint32_t rw_ = rcn < 0 ? 0xffffffffu : pword(fbits, rcn >> 7), lw_ = lcn < 0 ? 0xffffffffu : pword(fbits, lcn >> 7);
#define W_GO_R() do { x = rcn; RO::take(pR, vi, rcn, lcn); } while (0)
#define W_GO_L() do { x = lcn; RO::take(pL, vi, rcn, lcn); } while (0)
        if (!((vw_ >> (v & 31)) & 1u)) {
          pbit_set(vbits, v);
          if (!(vi & 1)) { W_EMIT(T_C); W_GO_R(); continue; }
        }
        const bool rvis = ((rw_ >> ((rcn >> 2) & 31)) & 1u) != 0, lvis = ((lw_ >> ((lcn >> 2) & 31)) & 1u) != 0;
        const int sym = rvis ? (lvis ? T_E : T_R) : (lvis ? T_L : T_S);
        W_EMIT(sym);
        if (sym == T_E) { sp--; break; }
        if (sym == T_R) { W_GO_L(); continue; }
        if (sym == T_L) { W_GO_R(); continue; }
        nsplit++;
        stack[sp - 1] = lcn; stack[sp] = rcn;
        sp++; top = rcn; top_known = true;
        break;
#undef W_GO_R
#undef W_GO_L
      }
    }
  }
#undef W_EMIT
  stg.tail_words(proc, nproc); stg.tail_bytes(symb, nproc);
  J.nsym = nproc; J.nsplit = nsplit; J.nstart = nstart; J.ninit = ninit;
  if (nproc + ninit != nf) J.status = -10;
  J.rb[0].n = (uint32_t)nstart;
  uint32_t z = 0; for (int i = 0; i < nstart; i++) z += J.start_bits[i] == 0;
  J.rb[0].zeros = z;
}


// ------------------------------------------------------------------------------------------------
// Cooperative-lane forms of the LDS walkers (the default whenever the bitmaps are in LDS).  A one-lane walk is bound by
// instruction issue, not by memory: a single wave issues about one instruction per 5 cycles, and a step of eb_walk_lane0 is
// ~100 instructions (two address computations and loads, three LDS reads behind branches, lane elections around the LDS
// atomics, scalar bookkeeping) = ~210 ns on top of the ~160 ns its dependent load costs (tools/latbench/seqbench).  Here the
// per-candidate work of a step is ONE vector instruction each: lane 0 handles the right neighbour, lane 1 the left one, every
// other lane the tip vertex - one load fetches both neighbours' records, one ds_read their two face-visited words and the
// vertex-visited word, one ballot turns the three tests into a scalar mask; the record the walk moves to is picked with
// v_readlane (lane select in an SGPR), so the step has no divergent branch and ~45 instructions.  The current face's bit is set with
// a plain LDS write (its word is known: a candidate lane read it one step earlier, or the pop test just did), outputs are staged in
// LDS (WalkStage) and flushed by all 64 lanes.  A second wave of the workgroup reads the walker's position from LDS and touches
// the 128-byte lines of the record table around it, so that the walker's loads hit in this CU's L1 / this XCD's L2 instead of
// paying an HBM miss per new line (seqbench: 157 -> 96 ns per dependent load on a strip-ordered table).
// Results are identical to eb_walk_lane0 / traverse_lane0 (same traversal, same output arrays).
// ------------------------------------------------------------------------------------------------
#define WALK_PUB_DWORDS 8                                // [0] walker position (corner code), [1] done flag
#define WALK_PF_LINES 64                                 // 128-byte lines the helper wave keeps touched around the walker
template <bool R8> struct CoopRec;
template <> struct CoopRec<true> {
  // two v_readlane, then scalar 64-bit shifts (written with 32-bit pieces the compiler moved the funnel shift back to the VALU)
  static __device__ __forceinline__ void take(const uvol_u2 &p, int sel, int &vi, int &rc, int &lc) {
    const unsigned long long q = ((unsigned long long)UVOL_READLANE(p.y, sel) << 32) | (unsigned long long)UVOL_READLANE(p.x, sel);
    vi = (int)((uint32_t)q & 0x1fffffu); rc = (int)((long long)(q << 22) >> 43); lc = (int)((long long)(q << 1) >> 43);
  }
};
template <> struct CoopRec<false> {
  static __device__ __forceinline__ void take(const uvol_i3 &p, int sel, int &vi, int &rc, int &lc) { vi = (int)UVOL_READLANE(p.x, sel); rc = (int)UVOL_READLANE(p.y, sel); lc = (int)UVOL_READLANE(p.z, sel); }
};
// record of `code` as wave-uniform scalars (every lane loads the same address: one request)
template <bool R8> __device__ __forceinline__ void coop_get(typename RecOps<R8>::Ptr rec, int code, int &vi, int &rc, int &lc) {
  int a, b, c; RecOps<R8>::get(rec, code, a, b, c); vi = UVOL_READFIRST(a); rc = UVOL_READFIRST(b); lc = UVOL_READFIRST(c);
}
// helper wave: keeps WALK_PF_LINES lines of the record table around the walker's published position touched
__device__ __forceinline__ void walk_prefetch_wave(const int32_t *rec_base, uint32_t rec_bytes, UVOL_L(uint32_t) pub, int shift /* corner code -> 128-byte line */) {
#ifndef HIPEMU
  const int lane = (int)(threadIdx.x & 63), nlines = (int)(rec_bytes >> 7);
  UVOL_G(const uint32_t) r = UVOL_TO_G(const uint32_t, reinterpret_cast<const uint32_t *>(rec_base));
  int base = -(1 << 30); uint32_t acc = 0;
  while (!__hip_atomic_load(&pub[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
    const int c = (int)__hip_atomic_load(&pub[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> shift;
    if (c < base + WALK_PF_LINES / 4 || c >= base + (3 * WALK_PF_LINES) / 4) {
      base = c - WALK_PF_LINES / 4;
      const int line = base + lane;
      if (line >= 0 && line < nlines) acc += r[32 * (size_t)line];
    }
    __builtin_amdgcn_s_sleep(2);
  }
  if (acc == 0x9e3779b9u) pub[2] = acc;               // keeps the loads alive
#endif
}

template <bool R8>
__device__ __forceinline__ void eb_walk_coop(GeoJob &J, UVOL_L(uint32_t) lds, uint32_t fw, UVOL_L(uint32_t) pub, int pf) {
  typedef RecOps<R8> RO;
  const int lane = (int)(threadIdx.x & 63);
  const bool cl = lane < 2;                              // candidate lanes: 0 = right neighbour, 1 = left neighbour; the others: tip vertex
  const int nf = (int)J.nf;
  const typename RO::Ptr rec = RO::ptr(J.rec[0]);
  UVOL_G(int32_t) proc = UVOL_TO_G(int32_t, J.proc); UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.stack);
  UVOL_G(int32_t) initc = UVOL_TO_G(int32_t, J.initc);
  UVOL_G(uint8_t) symb = UVOL_TO_G(uint8_t, J.symb); UVOL_G(uint8_t) start_bits = UVOL_TO_G(uint8_t, J.start_bits);
  UVOL_L(uint32_t) dummy = pub + 4 + (lane & 1);         // where the candidate lanes put the word the vertex lanes write back
  uint32_t pv = 0, sv = 0;                               // output staging: lane k = entry (nproc & ~63) + k of proc[] / symb[]
  int nproc = 0, ninit = 0, nstart = 0, nsplit = 0;
#define C_FWORD(k) ((uint32_t)UVOL_BCAST0(lds[k]))
  const bool rl = J.relabel != 0; UVOL_G(const int32_t) s_of_o = UVOL_TO_G(const int32_t, J.s_of_o);
  for (int fo = 0; fo < nf; fo++) {
    int f0 = fo;                                         // component starts follow the ORIGINAL face order (see eb_walk_lane0)
    if (rl) { if (nproc + ninit >= nf) break; f0 = UVOL_READFIRST(s_of_o[fo]); }
    else if ((fo & 31) == 0) { while (fo + 32 <= nf && C_FWORD(fo >> 5) == 0xffffffffu) fo += 32; if (fo >= nf) break; f0 = fo; }
    if ((C_FWORD(f0 >> 5) >> (f0 & 31)) & 1u) continue;
    int v0[3], r0_[3], l0_[3];
    for (int k = 0; k < 3; k++) coop_get<R8>(rec, 4 * f0 + k, v0[k], r0_[k], l0_[k]);
    const int o0[3] = { r0_[2], r0_[0], r0_[1] };
    int interior = 1, start = 4 * f0;
    for (int k = 0; k < 3; k++) {
      if (o0[k] < 0) { interior = 0; start = 4 * f0 + k; break; }
      if (v0[k] & 1) {
        int ci = 4 * f0 + k, rc = ci;
        while (rc >= 0) { ci = rc; int v_, r_, o; coop_get<R8>(rec, rc, v_, r_, o); rc = o < 0 ? -1 : code_prv(o); }
        interior = 0; start = code_prv(ci); break;
      }
    }
    if (lane == 0) start_bits[nstart] = (uint8_t)interior;
    nstart++;
    int from;
    if (interior) {
      for (int k = 0; k < 3; k++) { const int v = v0[k] >> 1; lds[fw + (v >> 5)] = (uint32_t)UVOL_BCAST0(lds[fw + (v >> 5)]) | (1u << (v & 31)); }
      lds[f0 >> 5] = C_FWORD(f0 >> 5) | (1u << (f0 & 31));
      if (lane == 0) initc[ninit] = 3 * f0 + 1;
      ninit++;
      from = o0[1];
      if (from < 0 || ((C_FWORD(from >> 7) >> ((from >> 2) & 31)) & 1u)) continue;
    } else from = start;
    int sp = 0;
    if (lane == 0) stack[sp] = from;
    sp++;
    int top = from; bool top_known = true;
    while (sp > 0) {
      int x;
      if (top_known) x = top; else { UVOL_WAVE_FENCE(); x = UVOL_BCAST0(stack[sp - 1]); }     // lane 0's own earlier store
      top_known = false;
      if (x < 0) { sp--; continue; }
      uint32_t xw = C_FWORD(x >> 7);                     // face-visited word of x's face
      if ((xw >> ((x >> 2) & 31)) & 1u) { sp--; continue; }
      int vi, rcn, lcn;
      coop_get<R8>(rec, x, vi, rcn, lcn);
      // One step = straight-line code with ONE taken branch (the back edge): a lone wave pays ~40 cycles of instruction fetch per
      // taken branch, so the common symbols (C, R, L) are resolved with scalar selects; S / E (a few % of the steps) and the
      // write-out of the staged outputs (every 64th step) leave the line.
      for (;;) {
        const int face = x >> 2;
        const int cand = lane == 0 ? rcn : lcn; const bool cvalid = cand >= 0;
        const int ccode = cvalid ? cand : x;
        const typename RO::Pre pre = RO::pre(rec, ccode);                        // lanes 0 / 1: the two records this step can move to
        if (pf) pub[0] = (uint32_t)x;                                            // for the prefetch wave
        lds[face >> 5] = xw | (1u << (face & 31));                               // face visited (plain write: xw is current)
        pv = UVOL_WRITELANE(3 * face + (x & 3), nproc & 63, pv);
        const int v = vi >> 1;
        const uint32_t widx = cl ? (uint32_t)ccode >> 7 : fw + (uint32_t)(v >> 5);
        const uint32_t sh = cl ? ((uint32_t)cand >> 2) & 31u : (uint32_t)v & 31u;
        const uint32_t word = lds[widx];
        const bool hit = ((word >> sh) & 1u) != 0 || (cl && !cvalid);
        const uint32_t m = (uint32_t)__ballot(hit) & 7u;                         // bit 0: right visited, 1: left visited, 2: tip vertex visited
        (cl ? dummy : lds + widx)[0] = word | (1u << sh);                        // the tip's bit (already set when it was visited)
        const bool ccase = (((m >> 2) | (uint32_t)vi) & 1u) == 0;                // tip unvisited and not on a boundary: C
        const uint32_t sym = ccase ? 0u : 1u + (m & 2u) + ((m & 1u) << 2);       // S = 1, L = 3 (left visited), R = 5 (right visited), E = 7
        sv = UVOL_WRITELANE(sym, nproc & 63, sv);
        nproc++;
        if (__builtin_expect((nproc & 63) == 0, 0)) { proc[nproc - 64 + lane] = (int32_t)pv; symb[nproc - 64 + lane] = (uint8_t)sv; }
        if (__builtin_expect((0x82u >> sym) & 1u, 0)) {                            // E (7) or S (1): the run of C / R / L steps ends
          if (sym == 7u) { sp--; break; }
          nsplit++;
          if (lane == 0) { stack[sp - 1] = lcn; stack[sp] = rcn; }
          sp++; top = rcn; top_known = true;
          break;
        }
        const int sel = (int)(sym >> 2);                                          // R (5): the walk goes left; C (0) and L (3): right
        x = sel ? lcn : rcn;
        xw = UVOL_READLANE(word, sel);
        CoopRec<R8>::take(pre, sel, vi, rcn, lcn);
      }
    }
  }
  if (lane < (nproc & 63)) { proc[(nproc & ~63) + lane] = (int32_t)pv; symb[(nproc & ~63) + lane] = (uint8_t)sv; }
  if (lane == 0) {
    J.nsym = nproc; J.nsplit = nsplit; J.nstart = nstart; J.ninit = ninit;
    if (nproc + ninit != nf) J.status = -10;
    J.rb[0].n = (uint32_t)nstart;
    UVOL_WAVE_FENCE();
    uint32_t z = 0; for (int i = 0; i < nstart; i++) z += J.start_bits[i] == 0;
    J.rb[0].zeros = z;
  }
}
#undef C_FWORD

// LDS: [face bits, fw words][vertex bits, vcap_words]; vcap_words is sized by the host from the input attribute counts and
// the LDS slot (a table with more vertices keeps its vertex bitmap in global memory).  A mesh whose face bitmap does not fit
// LDS is walked by the lane-per-walker kernels below (nothing in LDS).
template <bool R8>
__global__ void __launch_bounds__(128) k_eb_walk(GeoJob *jobs, int vcap_words, int pf) {
  GeoJob &J = jobs[blockIdx.x];
  UVOL_SERIAL_PRIO();
  UVOL_DYN_SMEM(uint32_t, lds);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool ok = J.status == 0;
  const uint32_t fw = ((uint32_t)J.nf + 31) / 32, vw = (J.nverts_t[0] + 31) / 32, vcw = (uint32_t)vcap_words;
  const bool v_in_lds = vw <= vcw;
  const uint32_t stg_off = (fw + vcw + 3u) & ~3u;          // WALK_STG_DWORDS of output staging + WALK_PUB_DWORDS behind the bitmaps
  if (ok) for (uint32_t k = tid; k < fw + vcw; k += 128) lds[k] = 0;
  if (tid < WALK_PUB_DWORDS) lds[stg_off + WALK_STG_DWORDS + tid] = 0;
  __syncthreads();
  if (!ok) return;
  UVOL_L(uint32_t) stg = UVOL_TO_L(uint32_t, lds) + stg_off; UVOL_L(uint32_t) pub = stg + WALK_STG_DWORDS;
  if (v_in_lds) {
    if (wave == 1) { if (pf) walk_prefetch_wave(J.rec[0], (uint32_t)((R8 ? 32 : 64) * (size_t)J.nf), pub, R8 ? 4 : 3); return; }
    eb_walk_coop<R8>(J, UVOL_TO_L(uint32_t, lds), fw, pub, pf);
    pub[1] = 1u;
    return;
  }
  if (tid != 0) return;
  eb_walk_lane0<R8>(J, UVOL_TO_L(uint32_t, lds), UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.vvis)), stg);
}

// face_time[f] = index of the symbol that encoded face f (-1 for the faces that only start a component): the inverse
// of proc[], built in parallel so that the serial walker has no scatter store in its loop
__global__ void __launch_bounds__(UVOL_BLOCK) k_face_time(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (i < (uint32_t)J.nsym) J.face_time[J.proc[i] / 3] = (int32_t)i;
}
// v2d[t][vertex] = position of the vertex in the coding order of table t: the inverse of order[t][] (same reason)
__global__ void __launch_bounds__(UVOL_BLOCK) k_v2d(GeoJob *jobs, int r8) {
  JOB_OR_RETURN;
  const int t = blockIdx.z;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (i >= J.ne[t]) return;
  if (t > 0 && (t - 1 >= J.nad || !J.interior_seams[t - 1])) return;
  const int c = J.order[t][i];
  const size_t code = (size_t)code_of_corner(c);
  int vi;
  if (r8 == 2) { const uint32_t *q = reinterpret_cast<const uint32_t *>(J.rec[1 + t]) + 4 * (size_t)(c / 3); vi = (int)((uint32_t)((((uint64_t)q[1] << 32) | q[0]) >> (21 * (c % 3))) & 0x1fffffu); }
  else vi = r8 ? (int)((uint32_t)J.rec[1 + t][2 * code] & 0x1fffffu) : J.rec[1 + t][4 * code];
  J.v2d[t][vi >> 1] = (int32_t)i;
}

// topology-split events (CheckAndStoreTopologySplitEvent): symbol i contributes an event for each already-encoded
// right / left neighbour whose own symbol is S.  flag[i] = number of events (0..2), compacted in symbol order.
__device__ inline int eb_events_of(const GeoJob &J, uint32_t i, int ev_spl[2], int ev_edge[2]) {
  const int sym = J.symb[i]; int n = 0;
  if (sym != 5 && sym != 3 && sym != 7) return 0;
  const int c = J.proc[i];
  const int rcn = J.opp[g_nxt(c)], lcn = J.opp[g_prv(c)];
  if ((sym == 5 || sym == 7) && rcn >= 0) { const int t = J.face_time[rcn / 3]; if (t >= 0 && J.symb[t] == 1) { ev_spl[n] = t; ev_edge[n] = 1; n++; } }
  if ((sym == 3 || sym == 7) && lcn >= 0) { const int t = J.face_time[lcn / 3]; if (t >= 0 && J.symb[t] == 1) { ev_spl[n] = t; ev_edge[n] = 0; n++; } }
  return n;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_eb_event_flags(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  uint32_t n = 0;
  if (i < J.nf) { int a[2], b[2]; n = i < (uint32_t)J.nsym ? (uint32_t)eb_events_of(J, i, a, b) : 0u; J.evcnt[i] = (uint8_t)n; }
  const uint32_t tot = block_sum(n);
  if (threadIdx.x == 0 && blockIdx.x < uvol_blocks_dev(J.nf)) J.bsum2[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_eb_event_compact(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  // events are rare (two per S symbol): a block whose scanned sums say 'none' has nothing to place (uniform: read by every thread
  // from the same two words)
  if (J.status == 0 && blockIdx.x != 0 && blockIdx.x < uvol_blocks_dev(J.nf) && J.bsum2[blockIdx.x + 1] == J.bsum2[blockIdx.x]) return;
  const bool live = J.status == 0 && i < J.nf;
  uint32_t v = live ? J.evcnt[i] : 0, tot;
  const uint32_t pos = block_excl_scan(v, &tot) + ((J.status == 0 && blockIdx.x <= uvol_blocks_dev(J.nf)) ? J.bsum2[blockIdx.x] : 0);
  if (live && v) {
    int a[2], b[2]; const int n = eb_events_of(J, i, a, b);
    for (int k = 0; k < n; k++) { J.ev_src[pos + k] = (int)i; J.ev_spl[pos + k] = a[k]; J.ev_edge[pos + k] = (uint8_t)b[k]; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) J.nev = (int)J.bsum2[uvol_blocks_dev(J.nf)];
}

// working copies the replay mutates: corner -> vertex map (S symbols re-map corners to new vertices) and the valence per vertex
__global__ void __launch_bounds__(UVOL_BLOCK) k_valence_init(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t t = blockIdx.x * UVOL_BLOCK + threadIdx.x, stride = gridDim.x * UVOL_BLOCK;
  { const uint32_t n4 = J.nc / 4;                                                     // 16 bytes per lane (both arrays are 16-byte aligned)
    const uint4 *src = reinterpret_cast<const uint4 *>(J.vert); uint4 *dst = reinterpret_cast<uint4 *>(J.c2vm);
    for (uint32_t q = t; q < n4; q += stride) dst[q] = src[q];
    for (uint32_t c = 4 * n4 + t; c < J.nc; c += stride) J.c2vm[c] = J.vert[c]; }
  const uint32_t nv0 = J.nverts_t[0] < J.ecap ? J.nverts_t[0] : J.ecap;
  for (uint32_t v = t; v < nv0; v += stride) J.vval[v] = J.ring_d[v];
}
// valence bookkeeping replay: ctx_of[i] = context (0..5) under which symbol i-1 is coded (i >= 1).
// The context of symbol i is the clamped valence of the vertex at next(corner_i) just before i updates it.  Between two
// split symbols valences only receive fixed decrements (C: n-1 p-1; R: a-1 n-1 p-2; L: a-1 n-2 p-1; E: a-2 n-2 p-2), so
// a run of up to 64 symbols is resolved by the whole wave at once: lane j reads the run-start valence of its vertex
// and subtracts what lanes k < j apply to that same vertex (one pass of v_readlane broadcasts), then every lane posts
// its three decrements with atomic adds.  Only an S symbol (vertex split: ring walks + corner re-mapping) is serial.
__global__ void __launch_bounds__(64) k_eb_valence(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.x];
  UVOL_SERIAL_PRIO();
  const uint32_t lane = threadIdx.x;
  const bool ok = J.status == 0;
  const int nsym = ok ? J.nsym : 0, nc = (int)J.nc;
  const int32_t *opp = J.opp, *proc = J.proc, *ftime = J.face_time; const uint8_t *symb = J.symb;
  int32_t *vval = J.vval, *c2vm = J.c2vm;
  // initial valences / corner->vertex replica: filled by k_valence_init (parallel) before this launch
  const int nv0 = ok ? (int)J.nverts_t[0] : 0;
  (void)nc;
  int nvval = nv0;
  for (int base = 0; base < nsym; base += 64) {
    const int mi = base + (int)lane;
    // lane-parallel gather of the chunk's corners, symbols and vertex ids (valid until an S re-maps corners)
    int c_ = 0, s_ = 0, va_ = 0, vn_ = 0, vp_ = 0;
    if (mi < nsym) { c_ = proc[mi]; s_ = symb[mi]; va_ = c2vm[c_]; vn_ = c2vm[g_nxt(c_)]; vp_ = c2vm[g_prv(c_)]; }
    const int cnt = nsym - base < 64 ? nsym - base : 64;
    // decrements of this lane's symbol, packed a | n << 2 | p << 4
    const uint32_t dpk = s_ == 0 ? 0x14u : (s_ == 5 ? 0x25u : (s_ == 3 ? 0x19u : 0x2au));
    int start = 0;
    while (start < cnt) {
      const unsigned long long smask = __ballot((int)lane >= start && (int)lane < cnt && s_ == 1);
      const int e = smask ? (int)(__ffsll((long long)smask) - 1) : cnt;          // first split symbol of [start, cnt)
      if (e > start) {                                                          // run [start, e) without a split
        const bool act = (int)lane >= start && (int)lane < e;
        const int v_start = act ? UVOL_ALOAD(&vval[vn_]) : 0;
        int acc = 0;
        for (int k = start; k + 1 < e; k++) {
          const int ka = (int)UVOL_READLANE(va_, k), kn = (int)UVOL_READLANE(vn_, k), kp = (int)UVOL_READLANE(vp_, k);
          const uint32_t kd = UVOL_READLANE(dpk, k);
          const int hit = (vn_ == ka ? (int)(kd & 3u) : 0) + (vn_ == kn ? (int)((kd >> 2) & 3u) : 0) + (vn_ == kp ? (int)(kd >> 4) : 0);
          acc += (int)lane > k ? hit : 0;
        }
        if (act) {
          const int av = v_start - acc;
          if (mi > 0) { const int cv = av < 2 ? 2 : (av > 7 ? 7 : av); J.ctx_of[mi] = (uint8_t)(cv - 2); }
          if (dpk & 3u) UVOL_AADD(&vval[va_], -(int)(dpk & 3u));
          UVOL_AADD(&vval[vn_], -(int)((dpk >> 2) & 3u));
          UVOL_AADD(&vval[vp_], -(int)(dpk >> 4));
        }
        UVOL_WAVE_FENCE();
        UVOL_WAVE_SYNC();
      }
      if (e < cnt) {                                                            // the split symbol: serial, lane 0
        const int i = base + e;
        const int lc = (int)UVOL_READLANE(c_, e);
        const int ia = (int)UVOL_READLANE(va_, e), in_ = (int)UVOL_READLANE(vn_, e), ip = (int)UVOL_READLANE(vp_, e);
        if (lane == 0) {
          const int nx = g_nxt(lc), pv = g_prv(lc);
          const int val_n = UVOL_ALOAD(&vval[in_]), val_p = UVOL_ALOAD(&vval[ip]);
          UVOL_ASTORE(&vval[in_], val_n - 1); UVOL_ASTORE(&vval[ip], val_p - 1);
          int nleft = 0, a = opp[pv];
          while (a >= 0) { if (ftime[a / 3] <= i) break; nleft++; a = opp[g_nxt(a)]; }
          UVOL_ASTORE(&vval[ia], nleft + 1);
          const int newv = nvval; int nright = 0; a = opp[nx];
          while (a >= 0) { if (ftime[a / 3] <= i) break; nright++; c2vm[g_nxt(a)] = newv; a = opp[g_prv(a)]; }
          UVOL_ASTORE(&vval[nvval], nright + 1);
          if (i > 0) { const int cv = val_n < 2 ? 2 : (val_n > 7 ? 7 : val_n); J.ctx_of[i] = (uint8_t)(cv - 2); }
        }
        nvval++;
        // refresh the not-yet-consumed vertex ids of this chunk (corners right of the split now map to the new vertex)
        UVOL_WAVE_FENCE();
        UVOL_WAVE_SYNC();
        if (mi < nsym && (int)lane > e) { va_ = c2vm[c_]; vn_ = c2vm[g_nxt(c_)]; vp_ = c2vm[g_prv(c_)]; }
      }
      start = e + 1;
    }
  }
}

// symbols -> the six valence-context streams, in symbol order (wave ballots give each symbol its slot)
__global__ void __launch_bounds__(64) k_eb_ctx(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.x];
  UVOL_SERIAL_PRIO();
  const uint32_t lane = threadIdx.x;
  const int nsym = J.status == 0 ? J.nsym : 0;
  uint32_t base_c[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int base = 1; base < nsym; base += 64) {
    const int i = base + (int)lane;
    const bool in = i < nsym;
    const int cx = in ? J.ctx_of[i] : 7;
    const int ps = in ? J.symb[i - 1] : 0;
    const uint32_t id = ps == 0 ? 0u : (ps == 1 ? 1u : (ps == 3 ? 2u : (ps == 5 ? 3u : 4u)));
    for (int c = 0; c < 6; c++) {
      const unsigned long long m = __ballot(in && cx == c);
      if (in && cx == c) J.ctx_sym[c][base_c[c] + (uint32_t)__popcll(m & lt)] = id;
      base_c[c] += (uint32_t)__popcll(m);
    }
  }
  if (lane == 0 && J.status == 0) for (int c = 0; c < 6; c++) { J.ctx_n[c] = base_c[c]; J.rs[c].n = base_c[c]; }
}

// renumber into decoder order (decoder corner 3f+k <-> rot^k(processed corner f))
__device__ __forceinline__ int renum_first_corner(const GeoJob &J, uint32_t f) { return (int)f < J.nsym ? J.proc[J.nsym - 1 - (int)f] : J.initc[(int)f - J.nsym]; }
__global__ void __launch_bounds__(UVOL_BLOCK) k_renumber_a(GeoJob *jobs) {
  JOB_OR_RETURN;
  uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f >= J.nf) return;
  const int c = renum_first_corner(J, f);
  const int o[3] = { c, g_nxt(c), g_prv(c) };
  for (int k = 0; k < 3; k++) J.new_of_old[o[k]] = (int)(3 * f + k);
}
// One thread per NEW face: the renumbered tables (opposite corners, value ids, vertices under the decoder's corner numbering),
// the attribute seams (MeshAttributeCornerTable::InitFromAttribute) with the seam-bit eligibility flags and their block sums,
// and the 'a seam touches this vertex' bits.  The renumbering maps whole faces (rotated), so everything a corner needs from
// its own face is in the thread's registers (three 12-byte loads per array from the OLD face) and the ids across an edge
// come from the old face of the opposite corner - 20 loads per face where the per-corner k_renumber_b + k_seams pair issued 60.
__global__ void __launch_bounds__(UVOL_BLOCK) k_renumber_seams(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x, nf = J.nf;
  const bool in = f < nf;
  __shared__ uint32_t ecnt[3];                                         // eligible corners per 256-corner block (three per 256 faces)
  if (threadIdx.x < 3) ecnt[threadIdx.x] = 0;
  __syncthreads();
  if (in) {
    const int c0 = renum_first_corner(J, f);
    const int fo = 3 * (c0 / 3), r0 = c0 - fo;                          // old face, rotation
    int opp_[3], P[3], U[3], Nn[3], V[3];
    { const uvol_s3 a = *reinterpret_cast<const uvol_s3 *>(J.opp + fo), b = *reinterpret_cast<const uvol_s3 *>(J.cp + fo), c = *reinterpret_cast<const uvol_s3 *>(J.cu + fo),
                    d = *reinterpret_cast<const uvol_s3 *>(J.cn + fo), e = *reinterpret_cast<const uvol_s3 *>(J.vert + fo);
      const int ao[3] = { a.x, a.y, a.z }, bo[3] = { b.x, b.y, b.z }, co[3] = { c.x, c.y, c.z }, dn[3] = { d.x, d.y, d.z }, ev[3] = { e.x, e.y, e.z };
      for (int k = 0; k < 3; k++) { const int j = (r0 + k) % 3; opp_[k] = ao[j]; P[k] = bo[j]; U[k] = co[j]; Nn[k] = dn[j]; V[k] = ev[j]; } }
    int no[3];
    for (int k = 0; k < 3; k++) no[k] = opp_[k] < 0 ? GEO_INV : J.new_of_old[opp_[k]];
    // ids across each edge: the two other corners of the opposite corner's OLD face
    int bu[3][2], bn[3][2];
    for (int k = 0; k < 3; k++) {
      const int oo = opp_[k] < 0 ? 0 : opp_[k];
      bu[k][0] = J.cu[g_prv(oo)]; bu[k][1] = J.cu[g_nxt(oo)]; bn[k][0] = J.cn[g_prv(oo)]; bn[k][1] = J.cn[g_nxt(oo)];
    }
    { uvol_s3 w; w.x = no[0]; w.y = no[1]; w.z = no[2]; *reinterpret_cast<uvol_s3 *>(J.nopp + 3 * (size_t)f) = w;
      w.x = P[0]; w.y = P[1]; w.z = P[2]; *reinterpret_cast<uvol_s3 *>(J.npid + 3 * (size_t)f) = w;
      w.x = U[0]; w.y = U[1]; w.z = U[2]; *reinterpret_cast<uvol_s3 *>(J.nuid + 3 * (size_t)f) = w;
      w.x = Nn[0]; w.y = Nn[1]; w.z = Nn[2]; *reinterpret_cast<uvol_s3 *>(J.nnid + 3 * (size_t)f) = w;
      w.x = V[0]; w.y = V[1]; w.z = V[2]; *reinterpret_cast<uvol_s3 *>(J.bvert + 3 * (size_t)f) = w; }
    for (int k = 0; k < 3; k++) {
      const uint32_t c = 3 * f + k; const bool e = no[k] >= 0 && (uint32_t)no[k] / 3 > f;
      J.elig[c] = e ? 1 : 0;
      if (e) atomicAdd(&ecnt[(3 * threadIdx.x + k) >> 8], 1u);
    }
    for (int i = 0; i < J.nad; i++) {
      const bool uvk = J.att_kind[i] == 0;
      bool any = false;
      for (int k = 0; k < 3; k++) {
        uint8_t sm = 1;
        if (opp_[k] >= 0) {
          const int a0 = uvk ? U[(k + 1) % 3] : Nn[(k + 1) % 3], a1 = uvk ? U[(k + 2) % 3] : Nn[(k + 2) % 3];
          const int b0 = uvk ? bu[k][0] : bn[k][0], b1 = uvk ? bu[k][1] : bn[k][1];
          sm = (a0 != b0 || a1 != b1) ? 1 : 0;
          if (sm) {                                                      // both ends of the edge get split
            any = true;
            const uint32_t va = (uint32_t)V[(k + 1) % 3], vb = (uint32_t)V[(k + 2) % 3];
            atomicOr(&J.vseam[i][va >> 5], 1u << (va & 31)); atomicOr(&J.vseam[i][vb >> 5], 1u << (vb & 31));
          }
        }
        J.seam[i][3 * (size_t)f + k] = sm;
      }
      if (any) J.interior_seams[i] = 1;
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) { const uint32_t b = 3 * blockIdx.x + threadIdx.x; if (b < uvol_blocks_dev(J.nc)) J.bsum[b] = ecnt[threadIdx.x]; }
}
// seam bits of the eligible corners, in corner order.  SB_E corners per thread (8-byte loads of the flags and of both seam
// arrays): with one corner per thread the kernel was 5 M workgroups per batch, each a chain of two byte loads and a scan
#define SB_E 8
__global__ void __launch_bounds__(UVOL_BLOCK) k_seam_bits(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  const bool ok = J.status == 0;
  const uint32_t nc = ok ? J.nc : 0u, c0 = (blockIdx.x * UVOL_BLOCK + threadIdx.x) * SB_E;
  unsigned long long e8 = 0;
  if (c0 + SB_E <= nc) e8 = *reinterpret_cast<const unsigned long long *>(J.elig + c0);
  else for (uint32_t k = 0; c0 + k < nc && k < SB_E; k++) e8 |= (unsigned long long)J.elig[c0 + k] << (8 * k);
  e8 &= 0x0101010101010101ull;
  uint32_t cnt = (uint32_t)__popcll(e8), tot;
  uint32_t pos = block_excl_scan(cnt, &tot) + ((ok && blockIdx.x * SB_E <= uvol_blocks_dev(J.nc)) ? J.bsum[blockIdx.x * SB_E] : 0);
  // zero counts: one global atomic per block and attribute
  __shared__ uint32_t zc[2];
  if (threadIdx.x < 2) zc[threadIdx.x] = 0;
  __syncthreads();
  if (cnt) for (int i = 0; i < J.nad; i++) {
    unsigned long long s8 = 0;
    if (c0 + SB_E <= nc) s8 = *reinterpret_cast<const unsigned long long *>(J.seam[i] + c0);
    else for (uint32_t k = 0; c0 + k < nc && k < SB_E; k++) s8 |= (unsigned long long)J.seam[i][c0 + k] << (8 * k);
    uint32_t p = pos, z = 0;
    for (int k = 0; k < SB_E; k++) if ((e8 >> (8 * k)) & 1ull) { const uint8_t sb = (uint8_t)(s8 >> (8 * k)); J.seam_bits[i][p++] = sb; z += sb ? 0u : 1u; }
    if (z) atomicAdd(&zc[i], z);
  }
  __syncthreads();
  if (threadIdx.x < 2 && zc[threadIdx.x]) atomicAdd(&J.rb[1 + threadIdx.x].zeros, zc[threadIdx.x]);
  if (blockIdx.x == 0 && threadIdx.x == 0 && ok) {
    uint32_t n = J.bsum[uvol_blocks_dev(J.nc)];
    J.n_elig = n; for (int i = 0; i < J.nad; i++) J.rb[1 + i].n = n;
  }
}

// attribute vertices of the vertices an interior seam touches (grid z = attribute slot): pass a gives every segment (maximal
// run of fan corners no seam / boundary separates) an id nverts_base + k at its left-most corner, pass b hands it to the other
// corners of the segment; all other corners keep their base vertex
__global__ void __launch_bounds__(UVOL_BLOCK) k_aseg_a(GeoJob *jobs) {
  JOB_OR_RETURN;
  const int i = (int)blockIdx.z;
  if (i >= J.nad || !J.interior_seams[i]) return;
  const uint32_t c0 = blockIdx.x * (UVOL_BLOCK * GEO_ILP) + threadIdx.x, nc = J.nc;
  int32_t v[GEO_ILP]; uint32_t w[GEO_ILP];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK; v[k] = c < nc ? J.bvert[c] : 0; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) w[k] = J.vseam[i][(uint32_t)v[k] >> 5];
  // left-most corner of its segment <=> the edge to its left is a seam or a boundary <=> seam[next(c)] (k_seams marks boundaries
  // too); fetched for every corner (a neighbouring byte) so that the rare seam vertices cost no divergent round trip
  uint8_t sl[GEO_ILP];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK; sl[k] = J.seam[i][g_nxt(c < nc ? c : 0u)]; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) {
    const uint32_t c = c0 + k * UVOL_BLOCK;
    if (c >= nc) continue;
    if (!((w[k] >> ((uint32_t)v[k] & 31)) & 1u)) { J.avert[i][c] = v[k]; continue; }
    if (sl[k]) J.avert[i][c] = (int32_t)(J.nverts_t[0] + atomicAdd(&J.nseg[i], 1u));
  }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_aseg_b(GeoJob *jobs) {
  JOB_OR_RETURN;
  const int i = (int)blockIdx.z;
  if (i >= J.nad || !J.interior_seams[i]) return;
  const uint32_t c0 = blockIdx.x * (UVOL_BLOCK * GEO_ILP) + threadIdx.x, nc = J.nc;
  if (c0 == 0) { const uint32_t tot = J.nverts_t[0] + J.nseg[i]; J.nverts_t[2 + i] = tot; if (tot > J.ecap) J.status = GEO_E_WS_OVERFLOW; }
  uint32_t v[GEO_ILP], w[GEO_ILP];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK; v[k] = c < nc ? (uint32_t)J.bvert[c] : 0u; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) w[k] = J.vseam[i][v[k] >> 5];
  GTab T; T.opp = J.nopp; T.seam = J.seam[i];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) {
    const uint32_t c = c0 + k * UVOL_BLOCK;
    if (c >= nc || !((w[k] >> (v[k] & 31)) & 1u)) continue;
    int l = (int)c; uint32_t guard = 0;
    for (;;) { const int nl = gt_swl(T, l); if (nl < 0) break; l = nl; if (++guard > nc) { J.status = -22; return; } }
    if (l != (int)c) J.avert[i][c] = J.avert[i][l];
  }
}

// ------------------------------------------------------------------------------------------------
// K5: DepthFirstTraverser — serial per (table, frame), one lane each.  t=0 base table, t=1,2 attribute tables.
// One record load per face (RecOps); visited faces / vertices are bitmaps in LDS; order[] is the only output stream
// (v2d[], its inverse, is rebuilt by k_v2d).  Same structure as eb_walk_lane0.
// ------------------------------------------------------------------------------------------------
template <bool R8, typename FB, typename VB>
__device__ __forceinline__ void traverse_lane0(GeoJob &J, int t, FB fbits, VB vbits, UVOL_L(uint32_t) stg_lds) {
  typedef RecOps<R8> RO;
  const int nf = (int)J.nf;
  const typename RO::Ptr rec = RO::ptr(J.rec[1 + t]);
  UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.t_stack[t]); UVOL_G(int32_t) order = UVOL_TO_G(int32_t, J.order[t]);
  const int dz = UVOL_LANE_ZERO();
  WalkStage stg; stg.init(stg_lds);
  int n = 0;
#define T_EMIT(C) do { stg.w[n & (WALK_STG - 1)] = (C); n++; if ((n & (WALK_STG - 1)) == 0) stg.flush_words(order, n); } while (0)
  for (int f = 0; f < nf; f++) {
    if ((f & 31) == 0) { while (f + 32 <= nf && pword(fbits, f >> 5) == 0xffffffffu) f += 32; if (f >= nf) break; }
    if (pbit_get(fbits, f)) continue;
    int x = 4 * f, sp = 0;
    stack[sp] = x;
    sp++;
    int top = x; bool top_known = true;
    { int vn, vp, r_, l_; RO::get(rec, x + 1, vn, r_, l_); RO::get(rec, x + 2, vp, r_, l_); vn >>= 1; vp >>= 1;
      if (!pbit_get(vbits, vn)) { pbit_set(vbits, vn); T_EMIT(3 * f + 1); }
      if (!pbit_get(vbits, vp)) { pbit_set(vbits, vp); T_EMIT(3 * f + 2); } }
    while (sp > 0) {
      x = top_known ? top : stack[sp - 1];
      top_known = false;
      if (x < 0 || pbit_get(fbits, x >> 2)) { sp--; continue; }
      int vi, rc, lc;
      RO::get(rec, x, vi, rc, lc);
      for (;;) {
        const int face = x >> 2;
        // both records this step can move to are requested now and taken (readfirstlane) only by the branch that goes there
        const typename RO::Pre pR = RO::pre(rec, (rc < 0 ? x : rc) + dz), pL = RO::pre(rec, (lc < 0 ? x : lc) + dz);
        pbit_set(fbits, face);
        const int v = vi >> 1;
        // the three bitmap words this step can need, read together (one LDS round trip)
        const uint32_t vw_ = pword(vbits, v >> 5);
        const uint32_t rw_ = rc < 0 ? 0xffffffffu : pword(fbits, rc >> 7), lw_ = lc < 0 ? 0xffffffffu : pword(fbits, lc >> 7);
#define T_GO_R() do { x = rc; RO::take(pR, vi, rc, lc); } while (0)
#define T_GO_L() do { x = lc; RO::take(pL, vi, rc, lc); } while (0)
        if (!((vw_ >> (v & 31)) & 1u)) {
          pbit_set(vbits, v); T_EMIT(3 * face + (x & 3));
          if (!(vi & 1)) { T_GO_R(); continue; }
        }
        const bool rvis = ((rw_ >> ((rc >> 2) & 31)) & 1u) != 0, lvis = ((lw_ >> ((lc >> 2) & 31)) & 1u) != 0;
        if (rvis) { if (lvis) { sp--; break; } T_GO_L(); }
        else { if (lvis) T_GO_R(); else { stack[sp - 1] = lc; stack[sp] = rc; sp++; top = rc; top_known = true; break; } }
#undef T_GO_R
#undef T_GO_L
      }
    }
  }
#undef T_EMIT
  stg.tail_words(order, n);
  J.ne[t] = (uint32_t)n;
  if (t == 0 && J.nverts != 0xffffffffu && (uint32_t)n != J.nverts) J.status = -11;      // (the decode path has no expected count)
}


// cooperative-lane form of traverse_lane0 (see eb_walk_coop): same traversal, same order[] stream
template <bool R8>
__device__ __forceinline__ void traverse_coop(GeoJob &J, int t, UVOL_L(uint32_t) lds, uint32_t fw, UVOL_L(uint32_t) pub, int pf) {
  typedef RecOps<R8> RO;
  const int lane = (int)(threadIdx.x & 63);
  const bool cl = lane < 2;
  const int nf = (int)J.nf;
  const typename RO::Ptr rec = RO::ptr(J.rec[1 + t]);
  UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.t_stack[t]); UVOL_G(int32_t) order = UVOL_TO_G(int32_t, J.order[t]);
  UVOL_L(uint32_t) dummy = pub + 4 + (lane & 1);
  uint32_t ov = 0;                                       // staged order[] entries: lane k = entry (n & ~63) + k
  int n = 0;
#define C_FWORD(k) ((uint32_t)UVOL_BCAST0(lds[k]))
#define C_EMIT(C) do { ov = UVOL_WRITELANE((C), n & 63, ov); n++; if ((n & 63) == 0) order[n - 64 + lane] = (int32_t)ov; } while (0)
  for (int f = 0; f < nf; f++) {
    if ((f & 31) == 0) { while (f + 32 <= nf && C_FWORD(f >> 5) == 0xffffffffu) f += 32; if (f >= nf) break; }
    if ((C_FWORD(f >> 5) >> (f & 31)) & 1u) continue;
    int x = 4 * f, sp = 0;
    if (lane == 0) stack[sp] = x;
    sp++;
    int top = x; bool top_known = true;
    { int vn, vp, r_, l_; coop_get<R8>(rec, x + 1, vn, r_, l_); coop_get<R8>(rec, x + 2, vp, r_, l_); vn >>= 1; vp >>= 1;
      uint32_t w = (uint32_t)UVOL_BCAST0(lds[fw + (vn >> 5)]);
      if (!((w >> (vn & 31)) & 1u)) { lds[fw + (vn >> 5)] = w | (1u << (vn & 31)); C_EMIT(3 * f + 1); }
      w = (uint32_t)UVOL_BCAST0(lds[fw + (vp >> 5)]);
      if (!((w >> (vp & 31)) & 1u)) { lds[fw + (vp >> 5)] = w | (1u << (vp & 31)); C_EMIT(3 * f + 2); } }
    while (sp > 0) {
      if (top_known) x = top; else { UVOL_WAVE_FENCE(); x = UVOL_BCAST0(stack[sp - 1]); }
      top_known = false;
      if (x < 0) { sp--; continue; }
      uint32_t xw = C_FWORD(x >> 7);
      if ((xw >> ((x >> 2) & 31)) & 1u) { sp--; continue; }
      int vi, rc, lc;
      coop_get<R8>(rec, x, vi, rc, lc);
      for (;;) {                                          // straight-line step, see eb_walk_coop
        const int face = x >> 2;
        const int cand = lane == 0 ? rc : lc; const bool cvalid = cand >= 0;
        const int ccode = cvalid ? cand : x;
        const typename RO::Pre pre = RO::pre(rec, ccode);
        if (pf) pub[0] = (uint32_t)x;
        lds[face >> 5] = xw | (1u << (face & 31));
        const int v = vi >> 1;
        const uint32_t widx = cl ? (uint32_t)ccode >> 7 : fw + (uint32_t)(v >> 5);
        const uint32_t sh = cl ? ((uint32_t)cand >> 2) & 31u : (uint32_t)v & 31u;
        const uint32_t word = lds[widx];
        const bool hit = ((word >> sh) & 1u) != 0 || (cl && !cvalid);
        const uint32_t m = (uint32_t)__ballot(hit) & 7u;
        (cl ? dummy : lds + widx)[0] = word | (1u << sh);
        // a vertex seen for the first time takes the next place in the order (the slot is simply overwritten otherwise)
        ov = UVOL_WRITELANE(3 * face + (x & 3), n & 63, ov);
        const int fresh = (int)((m >> 2) & 1u) ^ 1;
        n += fresh;
        if (__builtin_expect(fresh && (n & 63) == 0, 0)) order[n - 64 + lane] = (int32_t)ov;
        const bool ccase = (((m >> 2) | (uint32_t)vi) & 1u) == 0;
        const uint32_t k = ccase ? 0u : 1u + (m & 3u);   // 0: go right (new interior vertex); 1: fork; 2: right visited -> left; 3: left visited -> right; 4: dead end
        if (__builtin_expect((0x12u >> k) & 1u, 0)) {                              // fork (1) or dead end (4)
          if (k == 4u) { sp--; break; }
          if (lane == 0) { stack[sp - 1] = lc; stack[sp] = rc; }
          sp++; top = rc; top_known = true; break;
        }
        const int sel = k == 2u ? 1 : 0;
        x = sel ? lc : rc;
        xw = UVOL_READLANE(word, sel);
        CoopRec<R8>::take(pre, sel, vi, rc, lc);
      }
    }
  }
#undef C_EMIT
#undef C_FWORD
  if (lane < (n & 63)) order[(n & ~63) + lane] = (int32_t)ov;
  if (lane == 0) {
    J.ne[t] = (uint32_t)n;
    if (t == 0 && J.nverts != 0xffffffffu && (uint32_t)n != J.nverts) J.status = -11;
  }
}

template <bool R8>
__global__ void __launch_bounds__(128) k_traverse(GeoJob *jobs, int vcap_words, int dbg) {
  GeoJob &J = jobs[blockIdx.y];
  const int t = blockIdx.x;
  UVOL_SERIAL_PRIO();
  UVOL_DYN_SMEM(uint32_t, lds);
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  const int ai = t > 0 ? t - 1 : 0;
  const bool ok = J.status == 0 && !(t > 0 && (ai >= J.nad || !J.interior_seams[ai]));
  const uint32_t fw = ((uint32_t)J.nf + 31) / 32, vw = (J.nverts_t[1 + t] + 31) / 32, vcw = (uint32_t)vcap_words;
  const bool v_in_lds = vw <= vcw;
  const uint32_t stg_off = (fw + vcw + 3u) & ~3u;
  if (ok) for (uint32_t k = tid; k < fw + vcw; k += 128) lds[k] = 0;
  if (tid < WALK_PUB_DWORDS) lds[stg_off + WALK_STG_DWORDS + tid] = 0;
  __syncthreads();
  if (!ok) return;
  UVOL_L(uint32_t) stg = UVOL_TO_L(uint32_t, lds) + stg_off; UVOL_L(uint32_t) pub = stg + WALK_STG_DWORDS;
  if (v_in_lds) {
    if (wave == 1) { if (dbg & 2) walk_prefetch_wave(J.rec[1 + t], (uint32_t)((R8 ? 32 : 64) * (size_t)J.nf), pub, R8 ? 4 : 3); return; }
    traverse_coop<R8>(J, t, UVOL_TO_L(uint32_t, lds), fw, pub, dbg & 2);
    pub[1] = 1u;
    return;
  }
  if (tid != 0) return;
  traverse_lane0<R8>(J, t, UVOL_TO_L(uint32_t, lds), UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.t_vvis[t])), stg);
}

// ------------------------------------------------------------------------------------------------
// Lane-per-walker forms of K4 / K5.  The wave-per-walker kernels above keep one dependent-load chain per wave and their
// visited bitmaps in LDS, which caps a CU at 3 - 6 walkers.  Here every LANE walks its own frame (or table of a frame): plain
// SIMT code, divergent branches, nothing in LDS, so the number of chains in flight is bounded by frames in HBM, not by LDS.
// The face-visited flag lives in the 4th slot of the face's own record block (it arrives with the record prefetch: no face
// bitmap), the vertex-visited bitmap is a per-walker word array in global memory touched by its one lane only (plain
// load / OR / store: a thread always sees its own stores).  `W` = lanes used per wave: few walkers are spread over many
// waves (less branch serialisation per step), many walkers are packed up to 64 per wave.
// Results are identical to the wave-per-walker kernels (same traversal, same output arrays).
// ------------------------------------------------------------------------------------------------
// `rec` is a typed global pointer (UVOL_G): global_load / global_store with exactly counted waits.  Through generic pointers
// every access was a flat_* instruction followed by s_waitcnt vmcnt(0) lgkmcnt(0), i.e. each step also waited for its own
// stores and for the neighbour prefetches it had just issued.
#define S_REC(code, vi, rc, lc)                                                                                         \
  do {                                                                                                                  \
    if (R8) { const uvol_u2 q_ = *(UVOL_G(const uvol_u2))(rec + 2 * (size_t)(code)); rec8_dec(q_.x, q_.y, vi, rc, lc); }  \
    else { const uvol_i4 q_ = *(UVOL_G(const uvol_i4))(rec + 4 * (size_t)(code)); vi = q_.x; rc = q_.y; lc = q_.z; }      \
  } while (0)
#define S_FLAG(code) (rec[(R8 ? 2 : 4) * (size_t)((code) | 3)])
// raw prefetch of a neighbour's record + its face flag (decoded only by the branch that moves there)
#define S_PRE(code, a, b, c, fl)                                                                                        \
  do {                                                                                                                  \
    if (R8) { const uvol_u2 q_ = *(UVOL_G(const uvol_u2))(rec + 2 * (size_t)(code)); a = q_.x; b = q_.y; c = 0; }          \
    else { const uvol_i4 q_ = *(UVOL_G(const uvol_i4))(rec + 4 * (size_t)(code)); a = (uint32_t)q_.x; b = (uint32_t)q_.y; c = (uint32_t)q_.z; } \
    fl = S_FLAG(code);                                                                                                  \
  } while (0)
#define S_TAKE(a, b, c, vi, rc, lc) do { if (R8) rec8_dec(a, b, vi, rc, lc); else { vi = (int)(a); rc = (int)(b); lc = (int)(c); } } while (0)

// One step of a lane is the SAME straight-line code whatever its symbol (C / R / L / S differ only in predicated selects and
// two predicated stack stores), so the lanes of a wave do not serialise on their symbols: frames of a real sequence have
// different connectivity and walk different paths, and the earlier branch-per-symbol form ran 3 - 4 x slower on them than on
// the bench's lattice frames, whose walkers happen to move in lock step (tools/exp_r3e: 450 vs 136 ms per 2160 frames, equal
// with one lane per wave).  Only the rare events leave the line: a dead end (E: pop the stack, a dependent load) and the search
// for the next component.  The S symbol is "go right and push the left neighbour": its record is already prefetched.
template <bool R8>
__device__ inline void eb_walk_simt(GeoJob &J) {
  const int nf = (int)J.nf;
  UVOL_G(uint32_t) rec = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.rec[0]));
  UVOL_G(uint32_t) vbits = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.vvis));
  UVOL_G(int32_t) proc = UVOL_TO_G(int32_t, J.proc); UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.stack); UVOL_G(int32_t) initc = UVOL_TO_G(int32_t, J.initc);
  UVOL_G(uint8_t) symb = UVOL_TO_G(uint8_t, J.symb); UVOL_G(uint8_t) start_bits = UVOL_TO_G(uint8_t, J.start_bits);
  const bool rl = J.relabel != 0; UVOL_G(const int32_t) s_of_o = UVOL_TO_G(const int32_t, J.s_of_o);
  int nproc = 0, ninit = 0, nstart = 0, nsplit = 0;
  int fo = 0, sp = 0, x = -1, vi = 0, rcn = -1, lcn = -1;
  for (;;) {
    if (x < 0) {                                          // rare: a corner to go on from - the stack, else the next component
      bool finished = false;
      for (;;) {
        if (sp > 0) {
          const int c = stack[sp - 1];
          if (c < 0 || S_FLAG(c)) { sp--; continue; }
          x = c; S_REC(x, vi, rcn, lcn);
          break;
        }
        if (fo >= nf || nproc + ninit >= nf) { finished = true; break; }
        const int f0 = rl ? s_of_o[fo] : fo;             // component starts follow the ORIGINAL face order
        fo++;
        if (S_FLAG(4 * f0)) continue;
        int v0[3], r0_[3], l0_[3];
        for (int k = 0; k < 3; k++) S_REC(4 * f0 + k, v0[k], r0_[k], l0_[k]);
        const int o0[3] = { r0_[2], r0_[0], r0_[1] };                       // opposite(k) = right field of corner (k + 2) % 3
        int interior = 1, start = 4 * f0;
        for (int k = 0; k < 3; k++) {
          if (o0[k] < 0) { interior = 0; start = 4 * f0 + k; break; }
          if (v0[k] & 1) {                // boundary vertex: swing right to the boundary edge
            int ci = 4 * f0 + k, rc = ci;
            while (rc >= 0) { ci = rc; int v_, r_, l_; S_REC(rc, v_, r_, l_); rc = l_ < 0 ? -1 : code_prv(l_); }      // left field = opposite(prev): swing right
            interior = 0; start = code_prv(ci); break;
          }
        }
        start_bits[nstart] = (uint8_t)interior;
        nstart++;
        int from;
        if (interior) {
          for (int k = 0; k < 3; k++) { const int v = v0[k] >> 1; const uint32_t w = vbits[v >> 5]; vbits[v >> 5] = w | (1u << (v & 31)); }
          S_FLAG(4 * f0) = 1u;
          initc[ninit] = 3 * f0 + 1;
          ninit++;
          from = o0[1];
          if (from < 0 || S_FLAG(from)) continue;
        } else from = start;
        stack[0] = from; sp = 1;
      }
      if (finished) break;
    }
    // ---- the common step ----
    S_FLAG(x) = 1u;
    uint32_t ra, rb, rc_, rfl, la, lb, lc_, lfl;
    S_PRE(rcn < 0 ? x : rcn, ra, rb, rc_, rfl);
    S_PRE(lcn < 0 ? x : lcn, la, lb, lc_, lfl);
    proc[nproc] = 3 * (x >> 2) + (x & 3);
    const int v = vi >> 1;
    const uint32_t vw = vbits[v >> 5];
    vbits[v >> 5] = vw | (1u << (v & 31));                                  // (already set when the tip was visited)
    const uint32_t vvis = (vw >> (v & 31)) & 1u;
    const uint32_t rvis = (rcn < 0 || rfl != 0) ? 1u : 0u, lvis = (lcn < 0 || lfl != 0) ? 1u : 0u;
    const bool ccase = ((vvis | (uint32_t)vi) & 1u) == 0;                    // tip unvisited and not on a boundary
    const uint32_t sym = ccase ? 0u : 1u + 2u * lvis + 4u * rvis;           // C 0, S 1, L 3, R 5, E 7
    symb[nproc] = (uint8_t)sym;
    nproc++;
    if (sym == 1u) { stack[sp - 1] = lcn; stack[sp] = rcn; sp++; nsplit++; }   // S: the left neighbour waits on the stack, the walk goes right
    if (sym == 7u) { sp--; x = -1; }
    else {
      const bool go_l = sym == 5u;
      x = go_l ? lcn : rcn;
      const uint32_t qa = go_l ? la : ra, qb = go_l ? lb : rb, qc = go_l ? lc_ : rc_;
      S_TAKE(qa, qb, qc, vi, rcn, lcn);
    }
  }
  J.nsym = nproc; J.nsplit = nsplit; J.nstart = nstart; J.ninit = ninit;
  if (nproc + ninit != nf) J.status = -10;
  J.rb[0].n = (uint32_t)nstart;
  uint32_t z = 0; for (int i = 0; i < nstart; i++) z += start_bits[i] == 0;
  J.rb[0].zeros = z;
}
template <bool R8>
__global__ void __launch_bounds__(64) k_eb_walk_simt(GeoJob *jobs, int n, int W) {
  const int lane = (int)threadIdx.x;
  if (lane >= W) return;
  const int j = (int)blockIdx.x * W + lane;
  if (j >= n) return;
  GeoJob &J = jobs[j];
  if (J.status != 0) return;
  eb_walk_simt<R8>(J);
}

// attribute sequencing with the same straight-line step (see eb_walk_simt)
template <bool R8>
__device__ inline void traverse_simt(GeoJob &J, int t) {
  const int nf = (int)J.nf;
  UVOL_G(uint32_t) rec = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.rec[1 + t]));
  UVOL_G(uint32_t) vbits = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.t_vvis[t]));
  UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.t_stack[t]); UVOL_G(int32_t) order = UVOL_TO_G(int32_t, J.order[t]);
  int n = 0, nvis = 0, f = 0, sp = 0, x = -1, vi = 0, rc = -1, lc = -1;
  for (;;) {
    if (x < 0) {                                          // rare: the stack, else the next unvisited face starts a component
      bool finished = false;
      for (;;) {
        if (sp > 0) {
          const int c = stack[sp - 1];
          if (c < 0 || S_FLAG(c)) { sp--; continue; }
          x = c; S_REC(x, vi, rc, lc);
          break;
        }
        if (f >= nf || nvis >= nf) { finished = true; break; }
        const int f0 = f; f++;
        if (S_FLAG(4 * f0)) continue;
        stack[0] = 4 * f0; sp = 1;
        int vn, vp, r_, l_; S_REC(4 * f0 + 1, vn, r_, l_); S_REC(4 * f0 + 2, vp, r_, l_); vn >>= 1; vp >>= 1;
        uint32_t w = vbits[vn >> 5];
        if (!((w >> (vn & 31)) & 1u)) { vbits[vn >> 5] = w | (1u << (vn & 31)); order[n] = 3 * f0 + 1; n++; }
        w = vbits[vp >> 5];
        if (!((w >> (vp & 31)) & 1u)) { vbits[vp >> 5] = w | (1u << (vp & 31)); order[n] = 3 * f0 + 2; n++; }
      }
      if (finished) break;
    }
    S_FLAG(x) = 1u;
    nvis++;
    uint32_t ra, rb, rc_, rfl, la, lb, lc_, lfl;
    S_PRE(rc < 0 ? x : rc, ra, rb, rc_, rfl);
    S_PRE(lc < 0 ? x : lc, la, lb, lc_, lfl);
    const int v = vi >> 1;
    const uint32_t vw = vbits[v >> 5];
    vbits[v >> 5] = vw | (1u << (v & 31));
    const uint32_t vvis = (vw >> (v & 31)) & 1u;
    if (!vvis) { order[n] = 3 * (x >> 2) + (x & 3); n++; }                  // a vertex seen for the first time takes the next place
    const uint32_t rvis = (rc < 0 || rfl != 0) ? 1u : 0u, lvis = (lc < 0 || lfl != 0) ? 1u : 0u;
    const bool ccase = ((vvis | (uint32_t)vi) & 1u) == 0;
    const uint32_t k = ccase ? 0u : 1u + rvis + 2u * lvis;                  // 0 / 3: right; 2: left; 1: fork (right, left waits); 4: dead end
    if (k == 1u) { stack[sp - 1] = lc; stack[sp] = rc; sp++; }
    if (k == 4u) { sp--; x = -1; }
    else {
      const bool go_l = k == 2u;
      x = go_l ? lc : rc;
      const uint32_t qa = go_l ? la : ra, qb = go_l ? lb : rb, qc = go_l ? lc_ : rc_;
      S_TAKE(qa, qb, qc, vi, rc, lc);
    }
  }
  J.ne[t] = (uint32_t)n;
  if (t == 0 && J.nverts != 0xffffffffu && (uint32_t)n != J.nverts) J.status = -11;      // (the decode path has no expected count)
}
// walker id = table * n + frame: the lanes of a wave walk the same table of consecutive frames (similar lengths)
template <bool R8>
__global__ void __launch_bounds__(64) k_traverse_simt(GeoJob *jobs, int n, int W) {
  const int lane = (int)threadIdx.x;
  if (lane >= W) return;
  const int id = (int)blockIdx.x * W + lane;
  if (id >= 3 * n) return;
  const int t = id / n, j = id - t * n;                  // (the three tables of ONE frame in neighbouring lanes was measured slower: 347 vs 307 / 202 ms)
  GeoJob &J = jobs[j];
  const int ai = t > 0 ? t - 1 : 0;
  if (J.status != 0 || (t > 0 && (ai >= J.nad || !J.interior_seams[ai]))) return;
  traverse_simt<R8>(J, t);
}
#undef S_REC
#undef S_FLAG
#undef S_PRE
#undef S_TAKE

// ------------------------------------------------------------------------------------------------
// The same two lane-per-walker kernels on ONE 16-byte record per FACE (format 2, pack_face_records): the three vertex fields, the
// three opposite-corner codes and the face-visited flag.  A step still moves by corner codes (4 * face + k): the vertex of corner k
// is vertex field k, its right / left neighbours are opposite fields (k + 1) % 3 / (k + 2) % 3 of the same record.  Against the
// 8-byte corner records this halves the bytes the walkers fetch and write back (the flag dirties the line it is in), puts eight faces
// instead of four on a 128-byte line (more of a walker's dependent loads hit a line a neighbouring face already brought in), brings a
// candidate's record AND its visited flag in one load instead of two, and halves the record tables (4 x 3.2 MB less per frame in flight).
// Batches whose face count or id space does not fit the 21-bit fields keep the 16-byte corner records (geo_rec8).
// ------------------------------------------------------------------------------------------------
#ifdef HIPEMU
struct uvol_u4 { uint32_t x, y, z, w; };
#else
typedef uint32_t uvol_u4 __attribute__((ext_vector_type(4)));
#endif
__device__ __forceinline__ uvol_u4 f16_load(UVOL_G(uint32_t) rec, int face) { return *(UVOL_G(const uvol_u4))(rec + 4 * (size_t)face); }
// LDM = 1 (UVOL_WALK_LD=1, diagnostic): the walkers' loads as agent-scope atomics, which do not look the line up in the CU's L1.  Measured
// and NOT the default: traversal 234 against 206 ms per 1280 frames (tools/experiments/exp_r4u.sh) - the L1 hits are worth more than what a
// load behind the walker's own store into the same line waits for
template <int LDM> __device__ __forceinline__ uvol_u4 f16_ld(UVOL_G(uint32_t) rec, int face) {
  if (LDM == 0) return f16_load(rec, face);
  UVOL_G(uint64_t) p = (UVOL_G(uint64_t))(rec + 4 * (size_t)face);
  const uint64_t lo = UVOL_ALOAD(p), hi = UVOL_ALOAD(p + 1);
  uvol_u4 q; q.x = (uint32_t)lo; q.y = (uint32_t)(lo >> 32); q.z = (uint32_t)hi; q.w = (uint32_t)(hi >> 32);
  return q;
}
template <int LDM> __device__ __forceinline__ uint32_t w_ld(UVOL_G(uint32_t) p) { return LDM == 0 ? *p : UVOL_ALOAD(p); }
__device__ __forceinline__ void f16_dec(const uvol_u4 &q, int k, int &vi, int &rc, int &lc) {
  const uint64_t lo = (uint64_t)q.x | ((uint64_t)q.y << 32), hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
  const int s = 21 * k, sr = k == 2 ? 0 : s + 21, sl = k == 0 ? 42 : s - 21;
  vi = (int)((uint32_t)(lo >> s) & 0x1fffffu);
  rc = (int)((uint32_t)(hi >> sr) << 11) >> 11;            // 21-bit field, all ones = none
  lc = (int)((uint32_t)(hi >> sl) << 11) >> 11;
}
// FB = true: the face-visited flags are ONE BIT PER FACE in an array of their own (J.fvis / J.t_fvis[t], zeroed with the workspace head)
// instead of bit 63 of the record.  The record lines then stay clean: with the flag in the record every step dirtied the very line
// the next steps read their neighbours from (written back once per line: 9.6 MB per frame and table, and a store into a line makes
// the following loads of that line go back to L2).  A step knows its own face's word from the step before - the word it tested the
// face in as a candidate -, so marking is one plain store and testing the two candidates two 4-byte loads beside the record loads.
template <bool FB> struct F16Vis {
  UVOL_G(uint32_t) rec; UVOL_G(uint32_t) fb;
  // is face f visited?  q = its record (FB = false), w = its word of the bitmap (FB = true)
  __device__ __forceinline__ bool seen(const uvol_u4 &q, uint32_t w, int f) const { return FB ? ((w >> (f & 31)) & 1u) != 0 : (q.y >> 31) != 0; }
  __device__ __forceinline__ uint32_t word(int f) const { return FB ? fb[f >> 5] : 0u; }
  __device__ __forceinline__ void mark(int f, const uvol_u4 &q, uint32_t w) const { if (FB) fb[f >> 5] = w | (1u << (f & 31)); else rec[4 * (size_t)f + 1] = q.y | 0x80000000u; }
};
template <bool FB, int LDM>
__device__ inline void eb_walk_simt_f16(GeoJob &J) {
  const int nf = (int)J.nf;
  UVOL_G(uint32_t) rec = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.rec[0]));
  UVOL_G(uint32_t) vbits = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.vvis));
  UVOL_G(int32_t) proc = UVOL_TO_G(int32_t, J.proc); UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.stack); UVOL_G(int32_t) initc = UVOL_TO_G(int32_t, J.initc);
  UVOL_G(uint8_t) symb = UVOL_TO_G(uint8_t, J.symb); UVOL_G(uint8_t) start_bits = UVOL_TO_G(uint8_t, J.start_bits);
  const bool rl = J.relabel != 0; UVOL_G(const int32_t) s_of_o = UVOL_TO_G(const int32_t, J.s_of_o);
  F16Vis<FB> V; V.rec = rec; V.fb = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.fvis));
  int nproc = 0, ninit = 0, nstart = 0, nsplit = 0;
  int fo = 0, sp = 0, x = -1, vi = 0, rcn = -1, lcn = -1;
  uvol_u4 q; q.x = q.y = q.z = q.w = 0; uint32_t myw = 0;
  for (;;) {
    if (x < 0) {                                          // rare: a corner to go on from - the stack, else the next component
      bool finished = false;
      for (;;) {
        if (sp > 0) {
          const int c = stack[sp - 1];
          if (c < 0) { sp--; continue; }
          const uvol_u4 qq = f16_ld<LDM>(rec, c >> 2); const uint32_t ww = V.word(c >> 2);
          if (V.seen(qq, ww, c >> 2)) { sp--; continue; }
          x = c; q = qq; myw = ww; f16_dec(q, x & 3, vi, rcn, lcn);
          break;
        }
        if (fo >= nf || nproc + ninit >= nf) { finished = true; break; }
        const int f0 = rl ? s_of_o[fo] : fo;             // component starts follow the ORIGINAL face order
        fo++;
        const uvol_u4 q0 = f16_ld<LDM>(rec, f0); const uint32_t w0 = V.word(f0);
        if (V.seen(q0, w0, f0)) continue;
        int v0[3], r0_[3], l0_[3];
        for (int k = 0; k < 3; k++) f16_dec(q0, k, v0[k], r0_[k], l0_[k]);
        const int o0[3] = { r0_[2], r0_[0], r0_[1] };                       // opposite(k) = right field of corner (k + 2) % 3
        int interior = 1, start = 4 * f0;
        for (int k = 0; k < 3; k++) {
          if (o0[k] < 0) { interior = 0; start = 4 * f0 + k; break; }
          if (v0[k] & 1) {                // boundary vertex: swing right to the boundary edge
            int ci = 4 * f0 + k, rc = ci;
            while (rc >= 0) { ci = rc; int v_, r_, l_; f16_dec(f16_ld<LDM>(rec, rc >> 2), rc & 3, v_, r_, l_); rc = l_ < 0 ? -1 : code_prv(l_); }      // left field = opposite(prev): swing right
            interior = 0; start = code_prv(ci); break;
          }
        }
        start_bits[nstart] = (uint8_t)interior;
        nstart++;
        int from;
        if (interior) {
          for (int k = 0; k < 3; k++) { const int v = v0[k] >> 1; const uint32_t w = w_ld<LDM>(vbits + (v >> 5)); vbits[v >> 5] = w | (1u << (v & 31)); }
          V.mark(f0, q0, w0);
          initc[ninit] = 3 * f0 + 1;
          ninit++;
          from = o0[1];
          if (from < 0) continue;
          { const int ff = from >> 2; uint32_t wf = V.word(ff); if (FB && (ff >> 5) == (f0 >> 5)) wf |= 1u << (f0 & 31); if (V.seen(f16_ld<LDM>(rec, ff), wf, ff)) continue; }
        } else from = start;
        stack[0] = from; sp = 1;
      }
      if (finished) break;
    }
    // ---- the common step (straight-line, see eb_walk_simt) ----
    const int f = x >> 2, rf = (rcn < 0 ? x : rcn) >> 2, lf = (lcn < 0 ? x : lcn) >> 2;
    V.mark(f, q, myw);
    const uvol_u4 qr = f16_ld<LDM>(rec, rf), ql = f16_ld<LDM>(rec, lf);
    uint32_t wr = V.word(rf), wl = V.word(lf);
    if (FB) { const uint32_t mine = myw | (1u << (f & 31)); if ((rf >> 5) == (f >> 5)) wr = mine | wr; if ((lf >> 5) == (f >> 5)) wl = mine | wl; }      // (this step's own bit, whatever the load saw)
    proc[nproc] = 3 * f + (x & 3);
    const int v = vi >> 1;
    const uint32_t vw = w_ld<LDM>(vbits + (v >> 5));
    vbits[v >> 5] = vw | (1u << (v & 31));                                  // (already set when the tip was visited)
    const uint32_t vvis = (vw >> (v & 31)) & 1u;
    const uint32_t rvis = (rcn < 0 || V.seen(qr, wr, rf)) ? 1u : 0u, lvis = (lcn < 0 || V.seen(ql, wl, lf)) ? 1u : 0u;
    const bool ccase = ((vvis | (uint32_t)vi) & 1u) == 0;                    // tip unvisited and not on a boundary
    const uint32_t sym = ccase ? 0u : 1u + 2u * lvis + 4u * rvis;           // C 0, S 1, L 3, R 5, E 7
    symb[nproc] = (uint8_t)sym;
    nproc++;
    if (sym == 1u) { stack[sp - 1] = lcn; stack[sp] = rcn; sp++; nsplit++; }   // S: the left neighbour waits on the stack, the walk goes right
    if (sym == 7u) { sp--; x = -1; }
    else {
      const bool go_l = sym == 5u;
      x = go_l ? lcn : rcn;
      q.x = go_l ? ql.x : qr.x; q.y = go_l ? ql.y : qr.y; q.z = go_l ? ql.z : qr.z; q.w = go_l ? ql.w : qr.w; myw = go_l ? wl : wr;
      f16_dec(q, x & 3, vi, rcn, lcn);
    }
  }
  J.nsym = nproc; J.nsplit = nsplit; J.nstart = nstart; J.ninit = ninit;
  if (nproc + ninit != nf) J.status = -10;
  J.rb[0].n = (uint32_t)nstart;
  uint32_t z = 0; for (int i = 0; i < nstart; i++) z += start_bits[i] == 0;
  J.rb[0].zeros = z;
}
template <bool FB, int LDM>
__global__ void __launch_bounds__(64) k_eb_walk_simt_f16(GeoJob *jobs, int n, int W) {
  const int lane = (int)threadIdx.x;
  if (lane >= W) return;
  const int j = (int)blockIdx.x * W + lane;
  if (j >= n) return;
  GeoJob &J = jobs[j];
  if (J.status != 0) return;
  eb_walk_simt_f16<FB, LDM>(J);
}
template <bool FB, int LDM>
__device__ inline void traverse_simt_f16(GeoJob &J, int t) {
  const int nf = (int)J.nf;
  UVOL_G(uint32_t) rec = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.rec[1 + t]));
  UVOL_G(uint32_t) vbits = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.t_vvis[t]));
  UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.t_stack[t]); UVOL_G(int32_t) order = UVOL_TO_G(int32_t, J.order[t]);
  F16Vis<FB> V; V.rec = rec; V.fb = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.t_fvis[t]));
  int n = 0, nvis = 0, f = 0, sp = 0, x = -1, vi = 0, rc = -1, lc = -1;
  uvol_u4 q; q.x = q.y = q.z = q.w = 0; uint32_t myw = 0;
  for (;;) {
    if (x < 0) {                                          // rare: the stack, else the next unvisited face starts a component
      bool finished = false;
      for (;;) {
        if (sp > 0) {
          const int c = stack[sp - 1];
          if (c < 0) { sp--; continue; }
          const uvol_u4 qq = f16_ld<LDM>(rec, c >> 2); const uint32_t ww = V.word(c >> 2);
          if (V.seen(qq, ww, c >> 2)) { sp--; continue; }
          x = c; q = qq; myw = ww; f16_dec(q, x & 3, vi, rc, lc);
          break;
        }
        if (f >= nf || nvis >= nf) { finished = true; break; }
        // the next unvisited face in storage order.  A table with seams falls into many components (one per chart), and between two
        // of them this scan passes every face once: eight flag words per round trip instead of one record (a lane that scans holds up
        // the other walkers of its wave, and a walker alone spent a sixth of its time here)
        if (!FB) {
          int hit = -1;
          while (f < nf) {
            uint32_t y[8];
            for (int k = 0; k < 8; k++) { const int fk = f + k < nf ? f + k : nf - 1; y[k] = w_ld<LDM>(rec + 4 * (size_t)fk + 1); }
            uint32_t m = 0;
            for (int k = 0; k < 8; k++) m |= ((y[k] >> 31) ^ 1u) << k;
            if (nf - f < 8) m &= (1u << (nf - f)) - 1u;
            if (m) { hit = f + __builtin_ctz(m); break; }
            f += 8;
          }
          if (hit < 0) { finished = true; break; }
          f = hit;
        }
        const int f0 = f; f++;
        const uvol_u4 q0 = f16_ld<LDM>(rec, f0); const uint32_t w0 = V.word(f0);
        if (V.seen(q0, w0, f0)) continue;
        stack[0] = 4 * f0; sp = 1;
        int vn, vp, r_, l_; f16_dec(q0, 1, vn, r_, l_); f16_dec(q0, 2, vp, r_, l_); vn >>= 1; vp >>= 1;
        uint32_t w = w_ld<LDM>(vbits + (vn >> 5));
        if (!((w >> (vn & 31)) & 1u)) { vbits[vn >> 5] = w | (1u << (vn & 31)); order[n] = 3 * f0 + 1; n++; }
        w = w_ld<LDM>(vbits + (vp >> 5));
        if (!((w >> (vp & 31)) & 1u)) { vbits[vp >> 5] = w | (1u << (vp & 31)); order[n] = 3 * f0 + 2; n++; }
      }
      if (finished) break;
    }
    const int fc = x >> 2, rf = (rc < 0 ? x : rc) >> 2, lf = (lc < 0 ? x : lc) >> 2;
    V.mark(fc, q, myw);
    nvis++;
    const uvol_u4 qr = f16_ld<LDM>(rec, rf), ql = f16_ld<LDM>(rec, lf);
    uint32_t wr = V.word(rf), wl = V.word(lf);
    if (FB) { const uint32_t mine = myw | (1u << (fc & 31)); if ((rf >> 5) == (fc >> 5)) wr = mine | wr; if ((lf >> 5) == (fc >> 5)) wl = mine | wl; }
    const int v = vi >> 1;
    const uint32_t vw = w_ld<LDM>(vbits + (v >> 5));
    vbits[v >> 5] = vw | (1u << (v & 31));
    const uint32_t vvis = (vw >> (v & 31)) & 1u;
    if (!vvis) { order[n] = 3 * fc + (x & 3); n++; }                        // a vertex seen for the first time takes the next place
    const uint32_t rvis = (rc < 0 || V.seen(qr, wr, rf)) ? 1u : 0u, lvis = (lc < 0 || V.seen(ql, wl, lf)) ? 1u : 0u;
    const bool ccase = ((vvis | (uint32_t)vi) & 1u) == 0;
    const uint32_t k = ccase ? 0u : 1u + rvis + 2u * lvis;                  // 0 / 3: right; 2: left; 1: fork (right, left waits); 4: dead end
    if (k == 1u) { stack[sp - 1] = lc; stack[sp] = rc; sp++; }
    if (k == 4u) { sp--; x = -1; }
    else {
      const bool go_l = k == 2u;
      x = go_l ? lc : rc;
      q.x = go_l ? ql.x : qr.x; q.y = go_l ? ql.y : qr.y; q.z = go_l ? ql.z : qr.z; q.w = go_l ? ql.w : qr.w; myw = go_l ? wl : wr;
      f16_dec(q, x & 3, vi, rc, lc);
    }
  }
  J.ne[t] = (uint32_t)n;
  if (t == 0 && J.nverts != 0xffffffffu && (uint32_t)n != J.nverts) J.status = -11;      // (the decode path has no expected count)
}
template <bool FB, int LDM>
__global__ void __launch_bounds__(64) k_traverse_simt_f16(GeoJob *jobs, int n, int W) {
  const int lane = (int)threadIdx.x;
  if (lane >= W) return;
  const int id = (int)blockIdx.x * W + lane;
  if (id >= 3 * n) return;
  const int t = id / n, j = id - t * n;
  GeoJob &J = jobs[j];
  const int ai = t > 0 ? t - 1 : 0;
  if (J.status != 0 || (t > 0 && (ai >= J.nad || !J.interior_seams[ai]))) return;
  traverse_simt_f16<FB, LDM>(J, t);
}

// ------------------------------------------------------------------------------------------------
// K1: attribute min/max (orderable-float atomics) and quantisation of the entries in coding order
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(UVOL_BLOCK) k_minmax(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  uint32_t mn[5], mx[5];
  for (int k = 0; k < 5; k++) { mn[k] = 0xffffffffu; mx[k] = 0; }
  const bool ok = J.status == 0;
  // a few blocks per frame stride over the values: 10 atomics per BLOCK on the frame's bounding-box words (they were per wave)
  for (uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x; ok && i < J.n_pos; i += gridDim.x * UVOL_BLOCK)
    for (int k = 0; k < 3; k++) { const uint32_t u = g_float_order(J.pos[3 * (size_t)i + k]); mn[k] = u < mn[k] ? u : mn[k]; mx[k] = u > mx[k] ? u : mx[k]; }
  for (uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x; ok && J.has_uv && i < J.n_uv; i += gridDim.x * UVOL_BLOCK)
    for (int k = 0; k < 2; k++) { const uint32_t u = g_float_order(J.uv[2 * (size_t)i + k]); mn[3 + k] = u < mn[3 + k] ? u : mn[3 + k]; mx[3 + k] = u > mx[3 + k] ? u : mx[3 + k]; }
  __shared__ uint32_t smn[5], smx[5];
  if (threadIdx.x < 5) { smn[threadIdx.x] = 0xffffffffu; smx[threadIdx.x] = 0; }
  __syncthreads();
  for (int k = 0; k < 5; k++) {
    uint32_t a = mn[k], b = mx[k];
    for (int d = 32; d >= 1; d >>= 1) { uint32_t a2 = __shfl_xor(a, d), b2 = __shfl_xor(b, d); a = a2 < a ? a2 : a; b = b2 > b ? b2 : b; }
    if ((threadIdx.x & 63) == 0) { atomicMin(&smn[k], a); atomicMax(&smx[k], b); }
  }
  __syncthreads();
  if (threadIdx.x < 5 && ok && smn[threadIdx.x] <= smx[threadIdx.x]) {
    const int k = (int)threadIdx.x;
    if (k < 3) { atomicMin(&J.pos_min_u[k], smn[k]); atomicMax(&J.pos_max_u[k], smx[k]); }
    else if (J.has_uv) { atomicMin(&J.uv_min_u[k - 3], smn[k]); atomicMax(&J.uv_max_u[k - 3], smx[k]); }
  }
}
__device__ inline float quant_range(const uint32_t *mn, const uint32_t *mx, int ncomp) {
  float r = g_float_unorder(mx[0]) - g_float_unorder(mn[0]);
  for (int k = 1; k < ncomp; k++) { float d = g_float_unorder(mx[k]) - g_float_unorder(mn[k]); if (d > r) r = d; }
  if (r == 0.f) r = 1.f;
  return r;
}
__device__ inline void attr_order(const GeoJob &J, int i, const int32_t *&order, const int32_t *&v2d, const int32_t *&vert, uint32_t &ne) {
  if (J.interior_seams[i]) { order = J.order[1 + i]; v2d = J.v2d[1 + i]; vert = J.avert[i]; ne = J.ne[1 + i]; }
  else { order = J.order[0]; v2d = J.v2d[0]; vert = J.bvert; ne = J.ne[0]; }
}
__device__ inline void float_to_oct(const GOct &t, const float *v, int &s, int &tt) {
  double abs_sum = fabs((double)v[0]) + fabs((double)v[1]) + fabs((double)v[2]);
  double sv[3];
  if (abs_sum > 1e-6) { double sc = 1.0 / abs_sum; sv[0] = v[0] * sc; sv[1] = v[1] * sc; sv[2] = v[2] * sc; }
  else { sv[0] = 1; sv[1] = 0; sv[2] = 0; }
  int iv[3];
  iv[0] = (int)floor(sv[0] * t.CEN + 0.5);
  iv[1] = (int)floor(sv[1] * t.CEN + 0.5);
  iv[2] = t.CEN - g_iabs(iv[0]) - g_iabs(iv[1]);
  if (iv[2] < 0) { if (iv[1] > 0) iv[1] += iv[2]; else iv[1] -= iv[2]; iv[2] = 0; }
  if (sv[2] < 0) iv[2] *= -1;
  g_vec_to_oct(t, iv, s, tt);
}
// grid.z selects the attribute: 0 position, 1 uv, 2 normal
__global__ void __launch_bounds__(UVOL_BLOCK) k_quantize(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t p = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const int a = blockIdx.z;
  int lo = 0x7fffffff, hi = -0x7fffffff - 1; bool have = false;
  if (a == 0) {
    if (p < J.ne[0]) {
      const float range = quant_range(J.pos_min_u, J.pos_max_u, 3), inv = (float)((1u << J.qp) - 1) / range;
      const float *v = (J.relabel ? J.pos_s : J.pos) + 3 * (size_t)J.npid[J.order[0][p]];
      for (int k = 0; k < 3; k++) { float t = v[k] - g_float_unorder(J.pos_min_u[k]); t = t * inv; int q = (int)floorf(t + 0.5f); J.P[3 * p + k] = q; lo = q < lo ? q : lo; hi = q > hi ? q : hi; }
      have = true;
    }
  } else {
    int i = -1; for (int k = 0; k < J.nad; k++) if (J.att_kind[k] == a - 1) i = k;
    if (i >= 0) {
      const int32_t *order, *v2d, *vert; uint32_t ne; attr_order(J, i, order, v2d, vert, ne);
      if (p < ne) {
        if (a == 1) {
          const float range = quant_range(J.uv_min_u, J.uv_max_u, 2), inv = (float)((1u << J.qt) - 1) / range;
          const float *v = J.uv + 2 * (size_t)J.nuid[order[p]];
          for (int k = 0; k < 2; k++) { float t = v[k] - g_float_unorder(J.uv_min_u[k]); t = t * inv; int q = (int)floorf(t + 0.5f); J.U[2 * p + k] = q; lo = q < lo ? q : lo; hi = q > hi ? q : hi; }
          have = true;
        } else {
          GOct ot = g_oct(J.qn); int s, tt;
          float_to_oct(ot, J.nrm + 3 * (size_t)J.nnid[order[p]], s, tt);
          J.O[2 * p] = s; J.O[2 * p + 1] = tt;
        }
      }
    }
  }
  if (a < 2) {
    for (int d = 32; d >= 1; d >>= 1) { int l2 = __shfl_xor(lo, d), h2 = __shfl_xor(hi, d); lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
    unsigned long long any = __ballot(have);
    if ((threadIdx.x & 63) == 0 && any) { atomicMin(&J.wrap_lo[a], lo); atomicMax(&J.wrap_hi[a], hi); }
  }
}

// ------------------------------------------------------------------------------------------------
// K6: prediction residuals — parallel per entry (all originals are known on the encoder side)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(UVOL_BLOCK) k_pred_pos(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t p = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (p >= J.ne[0]) return;
  const int32_t *P = J.P, *v2d = J.v2d[0], *vert = J.bvert;
  long long pred[3] = {0, 0, 0};
  if (p > 0) {
    bool have = false;
    const int ci = J.order[0][p], oci = J.nopp[ci];
    if (oci >= 0) {
      const uint32_t a = (uint32_t)v2d[vert[oci]], bn = (uint32_t)v2d[vert[g_nxt(oci)]], bp = (uint32_t)v2d[vert[g_prv(oci)]];
      if (a < p && bn < p && bp < p) { for (int k = 0; k < 3; k++) pred[k] = (long long)P[3 * bn + k] + P[3 * bp + k] - P[3 * a + k]; have = true; }
    }
    if (!have) for (int k = 0; k < 3; k++) pred[k] = P[3 * (p - 1) + k];
  }
  for (int k = 0; k < 3; k++) J.sym_pos[3 * p + k] = g_sym_of(g_wrap_corr(J.wrap_lo[0], J.wrap_hi[0], P[3 * p + k], pred[k]));
}

__global__ void __launch_bounds__(UVOL_BLOCK) k_pred_uv(GeoJob *jobs) {
  JOB_OR_RETURN;
  int i = -1; for (int k = 0; k < J.nad; k++) if (J.att_kind[k] == 0) i = k;
  if (i < 0) return;
  const int32_t *order, *v2d, *vert; uint32_t ne; attr_order(J, i, order, v2d, vert, ne);
  const uint32_t p = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (p >= ne) return;
  const int32_t *U = J.U, *P = J.P, *bv2d = J.v2d[0], *bvert = J.bvert;
  const int c = order[p], cnx = g_nxt(c), cpv = g_prv(c);
  const uint32_t nd = (uint32_t)v2d[vert[cnx]], pd = (uint32_t)v2d[vert[cpv]];
  long long pred[2] = {0, 0}; bool have = false; uint8_t has_ori = 0, ori = 0;
  if (pd < p && nd < p) {
    const long long nuv[2] = { U[2 * nd], U[2 * nd + 1] }, puv[2] = { U[2 * pd], U[2 * pd + 1] };
    if (puv[0] == nuv[0] && puv[1] == nuv[1]) { pred[0] = puv[0]; pred[1] = puv[1]; have = true; }
    else {
      const int32_t *tip = P + 3 * bv2d[bvert[c]], *np_ = P + 3 * bv2d[bvert[cnx]], *pp_ = P + 3 * bv2d[bvert[cpv]];
      long long pn[3], pn2 = 0, dd = 0;
      for (int k = 0; k < 3; k++) { pn[k] = (long long)pp_[k] - np_[k]; pn2 += pn[k] * pn[k]; }
      if (pn2 != 0) {
        for (int k = 0; k < 3; k++) dd += pn[k] * ((long long)tip[k] - np_[k]);
        const long long pnuv[2] = { puv[0] - nuv[0], puv[1] - nuv[1] };
        const long long xuv[2] = { nuv[0] * pn2 + dd * pnuv[0], nuv[1] * pn2 + dd * pnuv[1] };
        long long cx2 = 0;
        for (int k = 0; k < 3; k++) { long long xp = np_[k] + (dd * pn[k]) / pn2; long long e = tip[k] - xp; cx2 += e * e; }
        const long long ns_ = (long long)g_isqrt((uint64_t)cx2 * (uint64_t)pn2);
        const long long cxuv[2] = { pnuv[1] * ns_, -pnuv[0] * ns_ };
        const long long p0[2] = { (xuv[0] + cxuv[0]) / pn2, (xuv[1] + cxuv[1]) / pn2 }, p1[2] = { (xuv[0] - cxuv[0]) / pn2, (xuv[1] - cxuv[1]) / pn2 };
        const long long cu0 = U[2 * p], cu1 = U[2 * p + 1];
        const long long d0 = (cu0 - p0[0]) * (cu0 - p0[0]) + (cu1 - p0[1]) * (cu1 - p0[1]);
        const long long d1 = (cu0 - p1[0]) * (cu0 - p1[0]) + (cu1 - p1[1]) * (cu1 - p1[1]);
        has_ori = 1;
        if (d0 < d1) { pred[0] = p0[0]; pred[1] = p0[1]; ori = 1; } else { pred[0] = p1[0]; pred[1] = p1[1]; ori = 0; }
        have = true;
      }
    }
  }
  if (!have) {
    if (nd < p) { pred[0] = U[2 * nd]; pred[1] = U[2 * nd + 1]; }
    else if (p > 0) { pred[0] = U[2 * (p - 1)]; pred[1] = U[2 * (p - 1) + 1]; }
  }
  J.has_ori[p] = has_ori; J.ori_val[p] = ori;
  for (int k = 0; k < 2; k++) J.sym_uv[2 * p + k] = g_sym_of(g_wrap_corr(J.wrap_lo[1], J.wrap_hi[1], U[2 * p + k], (long long)(int)pred[k]));
}
// orientation list in encoder push order (p descending); bit k = (o_k == o_{k-1}), o_{-1} = true
__global__ void __launch_bounds__(UVOL_BLOCK) k_ori_compact(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  const uint32_t p = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const uint32_t n = (J.status == 0 && J.has_uv) ? J.ne_uv : 0;
  uint32_t v = p < n ? J.has_ori[p] : 0, tot;
  uint32_t pos = block_excl_scan(v, &tot) + (blockIdx.x <= uvol_blocks_dev(n) ? J.bsum[blockIdx.x] : 0);
  if (p < n && v) J.ori_c[pos] = J.ori_val[p];
  if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) J.n_ori = J.bsum[uvol_blocks_dev(n)];
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_ori_bits(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t j = blockIdx.x * UVOL_BLOCK + threadIdx.x;   // list index (encoder push order)
  const uint32_t n = J.n_ori;
  if (j >= n) { if (j == 0) { J.rb[3].n = 0; } return; }
  const uint8_t o = J.ori_c[n - 1 - j], prev = j == 0 ? 1 : J.ori_c[n - j];
  const uint8_t bit = (o == prev) ? 1 : 0;
  J.ori_bits[j] = bit;
  if (!bit) atomicAdd(&J.rb[3].zeros, 1u);
  if (j == 0) J.rb[3].n = n;
}

// The geometric-normal predictor sums, over the faces around an entry's vertex, (a - cen) x (b - cen) of the face's quantised positions:
// the face's un-normalised normal, the same whichever of its corners the fan walk arrives at.  It is computed once per face here
// (9 position words through corner -> vertex -> coding order) instead of once per face AND vertex inside the walk, which then
// gathers one 24-byte normal per face instead of two positions through three dependent gathers each (k_pred_nrm: 53 -> 15.3 MB, k_face_normals itself 8.7 MB of
// HBM traffic per frame).
__global__ void __launch_bounds__(UVOL_BLOCK) k_face_normals(GeoJob *jobs) {
  JOB_OR_RETURN;
  if (!J.has_nrm) return;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f >= J.nf) return;
  const int32_t *P = J.P, *bv2d = J.v2d[0], *bvert = J.bvert;
  long long p[3][3];
  for (int k = 0; k < 3; k++) { const int32_t *q = P + 3 * (size_t)bv2d[bvert[3 * f + k]]; p[k][0] = q[0]; p[k][1] = q[1]; p[k][2] = q[2]; }
  long long dn[3], dp[3];
  for (int k = 0; k < 3; k++) { dn[k] = p[1][k] - p[0][k]; dp[k] = p[2][k] - p[0][k]; }
  const long long n0 = dn[1] * dp[2] - dn[2] * dp[1], n1 = dn[2] * dp[0] - dn[0] * dp[2], n2 = dn[0] * dp[1] - dn[1] * dp[0];
  // |components| < 2^(2 qp + 1): three 32-bit words per face up to 15 bits of quantisation (12 bytes per face), 64-bit words for 16
  if (J.qp <= 15) { int32_t *o = reinterpret_cast<int32_t *>(J.fnorm) + 3 * (size_t)f; o[0] = (int32_t)n0; o[1] = (int32_t)n1; o[2] = (int32_t)n2; }
  else { long long *o = J.fnorm + 3 * (size_t)f; o[0] = n0; o[1] = n1; o[2] = n2; }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_pred_nrm(GeoJob *jobs) {
  JOB_OR_RETURN;
  int i = -1; for (int k = 0; k < J.nad; k++) if (J.att_kind[k] == 1) i = k;
  if (i < 0) return;
  const int32_t *order, *v2d, *vert; uint32_t ne; attr_order(J, i, order, v2d, vert, ne);
  const uint32_t d = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (d == 0) J.rb[4].n = ne;
  if (d >= ne) return;
  GTab X; X.opp = J.nopp; X.seam = J.interior_seams[i] ? J.seam[i] : nullptr;
  const long long *FN = J.fnorm;
  const GOct ot = g_oct(J.qn);
  const int c0 = order[d];
  long long N[3] = {0, 0, 0};
  int c = c0; bool left = true; uint32_t guard = 0;
  while (c >= 0 && guard++ <= J.nc) {
    // (the face's normal, whichever corner of it c is: k_face_normals)
    if (J.qp <= 15) { const int32_t *fn = reinterpret_cast<const int32_t *>(FN) + 3 * (size_t)(c / 3); N[0] += fn[0]; N[1] += fn[1]; N[2] += fn[2]; }
    else { const long long *fn = FN + 3 * (size_t)(c / 3); N[0] += fn[0]; N[1] += fn[1]; N[2] += fn[2]; }
    if (left) { c = gt_swl(X, c); if (c == c0) break; if (c < 0) { left = false; c = gt_swr(X, c0); } }
    else c = gt_swr(X, c);
  }
  long long s = g_labs(N[0]) + g_labs(N[1]) + g_labs(N[2]);
  if (s > (1 << 29)) { long long qd = s / (1 << 29); for (int k = 0; k < 3; k++) N[k] /= qd; s = g_labs(N[0]) + g_labs(N[1]) + g_labs(N[2]); }
  int pv[3];
  if (s == 0) { pv[0] = ot.CEN; pv[1] = 0; pv[2] = 0; }
  else {
    long long aa = (N[0] * ot.CEN) / s, bb = (N[1] * ot.CEN) / s, cc = ot.CEN - g_labs(aa) - g_labs(bb);
    if (N[2] < 0) cc = -cc;
    pv[0] = (int)aa; pv[1] = (int)bb; pv[2] = (int)cc;
  }
  int ppos[2], pneg[2], cpos[2], cneg[2];
  g_vec_to_oct(ot, pv, ppos[0], ppos[1]);
  pv[0] = -pv[0]; pv[1] = -pv[1]; pv[2] = -pv[2];
  g_vec_to_oct(ot, pv, pneg[0], pneg[1]);
  const int orig[2] = { J.O[2 * d], J.O[2 * d + 1] };
  g_oct_corr(ot, orig, ppos, cpos); g_oct_corr(ot, orig, pneg, cneg);
  for (int k = 0; k < 2; k++) { cpos[k] = g_modmax(ot, cpos[k]); cneg[k] = g_modmax(ot, cneg[k]); }
  const int *ch; uint8_t flip;
  if (g_iabs(cpos[0]) + g_iabs(cpos[1]) < g_iabs(cneg[0]) + g_iabs(cneg[1])) { flip = 0; ch = cpos; } else { flip = 1; ch = cneg; }
  J.flips[d] = flip;
  if (!flip) atomicAdd(&J.rb[4].zeros, 1u);
  for (int k = 0; k < 2; k++) J.sym_nrm[2 * d + k] = (uint32_t)(ch[k] < 0 ? ch[k] + ot.MAXQ : ch[k]);
}

// ------------------------------------------------------------------------------------------------
// Sequential connectivity (DRACO_COMPRESSION_LEVEL 0: what stock `draco_encoder -cl 0` selects; north_star "edgebreaker /
// sequential connectivity").  No traversal at all, so every stage is parallel: points = the distinct (position, uv, normal)
// value triples in order of first appearance over the corners (hash table: first corner of every (pos, uv) pair, then of every
// (pair, normal) pair; flag scan), the index section = the point of every corner in the smallest storage type, every attribute
// coded per point with the DIFFERENCE predictor (previous point; wrap / canonicalised-octahedron transform) through the same
// histogram / table / rANS kernels as the edgebreaker path.  Every face is kept, also degenerate ones.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long sq_key(const GeoJob &J, uint32_t c, int level) {
  if (level == 0) return ((unsigned long long)J.canon[0][J.ipos[c]] << 32) | (unsigned long long)(J.has_uv ? J.canon[1][J.iuv[c]] : 0u);
  return ((unsigned long long)(uint32_t)J.sq_pu[c] << 32) | (unsigned long long)(J.has_nrm ? J.canon[2][J.inrm[c]] : 0u);
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_sq_clear(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  for (uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x; i < J.sq_cap; i += gridDim.x * UVOL_BLOCK) { J.sq_keys[i] = ~0ull; J.sq_val[i] = 0xffffffffu; }
}
// phase 0: validate the corner's indices (first level only), claim a slot for its key, keep the lowest corner; phase 1: read it back
__global__ void __launch_bounds__(UVOL_BLOCK) k_sq_hash(GeoJob *jobs, int level, int phase) {
  JOB_OR_RETURN;
  const uint32_t c = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (c >= 3 * J.nf_in) return;
  if (level == 0 && phase == 0 && (J.ipos[c] >= J.n_pos || (J.has_uv && J.iuv[c] >= J.n_uv) || (J.has_nrm && J.inrm[c] >= J.n_nrm))) { J.status = -2; return; }
  const unsigned long long key = sq_key(J, c, level);
  uint32_t s = (uint32_t)g_mix64(key) & (J.sq_cap - 1);
  for (uint32_t guard = 0; guard <= J.sq_cap; guard++) {
    unsigned long long cur = J.sq_keys[s];
    if (phase == 0 && cur == ~0ull) { const unsigned long long old = atomicCAS(&J.sq_keys[s], ~0ull, key); cur = old == ~0ull ? key : old; }
    if (cur == key) { if (phase == 0) atomicMin(&J.sq_val[s], c); else (level == 0 ? J.sq_pu : J.sq_first)[c] = (int32_t)J.sq_val[s]; return; }
    if (cur == ~0ull) break;
    s = (s + 1) & (J.sq_cap - 1);
  }
  J.status = -20;
}
// step 0: flag the first corner of every point (+ block sums); step 1 (after the scan): point ids of the first corners, corner of
// every point, the point count; step 2: every corner's point id + the byte count of its index; step 3 (after the second scan): bytes
__global__ void __launch_bounds__(UVOL_BLOCK) k_sq_points(GeoJob *jobs, int step) {
  GeoJob &J = jobs[blockIdx.y];
  const uint32_t c = blockIdx.x * UVOL_BLOCK + threadIdx.x, nc = 3 * J.nf_in;
  const bool live = J.status == 0 && c < nc;
  if (step == 0 || step == 2) {
    uint32_t v = 0;
    if (live && step == 0) v = J.sq_first[c] == (int32_t)c ? 1u : 0u;
    if (live && step == 2) {
      const uint32_t p = (uint32_t)J.sq_pid[J.sq_first[c]], np = J.sq_np; J.sq_pid[c] = (int32_t)p;
      v = np < 256u ? 1u : (np < (1u << 16) ? 2u : (np < (1u << 21) ? (p < 128u ? 1u : (p < 16384u ? 2u : 3u)) : 4u));
    }
    if (live) J.sq_flag[c] = (uint8_t)v;
    const uint32_t tot = block_sum(v);
    if (threadIdx.x == 0 && blockIdx.x < uvol_blocks_dev(nc)) J.bsum[blockIdx.x] = tot;
    return;
  }
  uint32_t v = live ? J.sq_flag[c] : 0, tot;
  const uint32_t pos = block_excl_scan(v, &tot) + ((J.status == 0 && blockIdx.x <= uvol_blocks_dev(nc)) ? J.bsum[blockIdx.x] : 0);
  if (step == 1) {
    if (live && v) { J.sq_pid[c] = (int32_t)pos; if (pos < J.ecap) J.sq_cop[pos] = (int32_t)c; }
    if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) {
      const uint32_t np = J.bsum[uvol_blocks_dev(nc)];
      J.sq_np = np; J.nf = J.nf_in; J.nc = nc; J.nverts = np; J.ne[0] = np;
      if (np > J.ecap) J.status = GEO_E_WS_OVERFLOW;
      J.rs[6].n = 3 * np; J.rs[7].n = J.has_uv ? 2 * np : 0; J.rs[8].n = J.has_nrm ? 2 * np : 0;
    }
  } else {
    if (live) {
      const uint32_t p = (uint32_t)J.sq_pid[c], np = J.sq_np; uint8_t *o = J.sq_idx + pos;
      if (np < 256u) o[0] = (uint8_t)p;
      else if (np < (1u << 16)) { o[0] = (uint8_t)p; o[1] = (uint8_t)(p >> 8); }
      else if (np < (1u << 21)) { uint32_t q = p; uint32_t k = 0; while (q >= 0x80u) { o[k++] = (uint8_t)(q | 0x80u); q >>= 7; } o[k] = (uint8_t)q; }
      else { o[0] = (uint8_t)p; o[1] = (uint8_t)(p >> 8); o[2] = (uint8_t)(p >> 16); o[3] = (uint8_t)(p >> 24); }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) J.sq_idx_bytes = J.bsum[uvol_blocks_dev(nc)];
  }
}
// per point: quantised values of its first corner (z: 0 position, 1 uv, 2 normal) + the wrap bounds
__global__ void __launch_bounds__(UVOL_BLOCK) k_sq_quant(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t p = blockIdx.x * UVOL_BLOCK + threadIdx.x; const int a = blockIdx.z;
  int lo = 0x7fffffff, hi = -0x7fffffff - 1; bool have = false;
  if (p < J.sq_np && !(a == 1 && !J.has_uv) && !(a == 2 && !J.has_nrm)) {
    const uint32_t c = (uint32_t)J.sq_cop[p];
    if (a == 0) {
      const float range = quant_range(J.pos_min_u, J.pos_max_u, 3), inv = (float)((1u << J.qp) - 1) / range;
      const float *v = J.pos + 3 * (size_t)J.canon[0][J.ipos[c]];
      for (int k = 0; k < 3; k++) { float t = v[k] - g_float_unorder(J.pos_min_u[k]); t = t * inv; const int q = (int)floorf(t + 0.5f); J.P[3 * p + k] = q; lo = q < lo ? q : lo; hi = q > hi ? q : hi; }
      have = true;
    } else if (a == 1) {
      const float range = quant_range(J.uv_min_u, J.uv_max_u, 2), inv = (float)((1u << J.qt) - 1) / range;
      const float *v = J.uv + 2 * (size_t)J.canon[1][J.iuv[c]];
      for (int k = 0; k < 2; k++) { float t = v[k] - g_float_unorder(J.uv_min_u[k]); t = t * inv; const int q = (int)floorf(t + 0.5f); J.U[2 * p + k] = q; lo = q < lo ? q : lo; hi = q > hi ? q : hi; }
      have = true;
    } else { GOct ot = g_oct(J.qn); int s_, t_; float_to_oct(ot, J.nrm + 3 * (size_t)J.canon[2][J.inrm[c]], s_, t_); J.O[2 * p] = s_; J.O[2 * p + 1] = t_; }
  }
  if (a < 2) {
    for (int d = 32; d >= 1; d >>= 1) { const int l2 = __shfl_xor(lo, d), h2 = __shfl_xor(hi, d); lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
    const unsigned long long any = __ballot(have);
    if ((threadIdx.x & 63) == 0 && any) { atomicMin(&J.wrap_lo[a], lo); atomicMax(&J.wrap_hi[a], hi); }
  }
}
// DIFFERENCE predictor: the previous point's value (zeros for the first point) -> symbols
__global__ void __launch_bounds__(UVOL_BLOCK) k_sq_pred(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t p = blockIdx.x * UVOL_BLOCK + threadIdx.x; const int a = blockIdx.z;
  if (p >= J.sq_np || (a == 1 && !J.has_uv) || (a == 2 && !J.has_nrm)) return;
  if (a == 0) { for (int k = 0; k < 3; k++) J.sym_pos[3 * p + k] = g_sym_of(g_wrap_corr(J.wrap_lo[0], J.wrap_hi[0], J.P[3 * p + k], p ? (long long)J.P[3 * (p - 1) + k] : 0ll)); }
  else if (a == 1) { for (int k = 0; k < 2; k++) J.sym_uv[2 * p + k] = g_sym_of(g_wrap_corr(J.wrap_lo[1], J.wrap_hi[1], J.U[2 * p + k], p ? (long long)J.U[2 * (p - 1) + k] : 0ll)); }
  else {
    const GOct ot = g_oct(J.qn);
    const int orig[2] = { J.O[2 * p], J.O[2 * p + 1] }, pred[2] = { p ? J.O[2 * (p - 1)] : 0, p ? J.O[2 * (p - 1) + 1] : 0 }; int corr[2];
    g_oct_corr(ot, orig, pred, corr);
    J.sym_nrm[2 * p] = (uint32_t)corr[0]; J.sym_nrm[2 * p + 1] = (uint32_t)corr[1];
  }
}

// single thread per frame: publish stream lengths once the entry counts are known
__global__ void __launch_bounds__(64) k_stream_setup(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.x];
  if (threadIdx.x != 0 || J.status != 0) return;
  J.rs[6].n = 3 * J.ne[0];
  J.rs[7].n = 0; J.rs[8].n = 0; J.ne_uv = 0; J.ne_nrm = 0;
  for (int i = 0; i < J.nad; i++) {
    uint32_t ne = J.interior_seams[i] ? J.ne[1 + i] : J.ne[0];
    if (J.att_kind[i] == 0) { J.rs[7].n = 2 * ne; J.ne_uv = ne; } else { J.rs[8].n = 2 * ne; J.ne_nrm = ne; }
  }
}

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

// ------------------------------------------------------------------------------------------------
// layout: small header pieces + piece list (single lane per frame), then a parallel gather
// ------------------------------------------------------------------------------------------------
__device__ inline void add_piece(GeoJob &J, const uint8_t *p, uint32_t len, uint32_t &total) {
  if (J.n_pieces >= GEO_MAXPIECES) { J.status = -40; return; }
  J.piece_ptr[J.n_pieces] = p; J.piece_len[J.n_pieces] = len; J.piece_off[J.n_pieces] = total; J.n_pieces++; total += len;
}
__device__ inline void put_i32(uint8_t *a, uint32_t &o, int32_t v) { for (int k = 0; k < 4; k++) a[o++] = (uint8_t)((uint32_t)v >> (8 * k)); }
__device__ inline void put_f32(uint8_t *a, uint32_t &o, float f) { uint32_t u; memcpy(&u, &f, 4); for (int k = 0; k < 4; k++) a[o++] = (uint8_t)(u >> (8 * k)); }
__device__ inline void add_rans(GeoJob &J, int s, uint32_t &total) {
  RansStream &S = J.rs[s];
  add_piece(J, S.head, S.head_len, total); add_piece(J, S.pay + S.pay_off, S.pay_len, total);
}
__global__ void __launch_bounds__(64) k_layout(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.x];
  if (threadIdx.x != 0 || J.status != 0) return;
  uint8_t *a = J.arena; uint32_t o = 0, total = 0, b0;
  J.n_pieces = 0;
  // header + connectivity header
  b0 = o;
  a[o++] = 'D'; a[o++] = 'R'; a[o++] = 'A'; a[o++] = 'C'; a[o++] = 'O'; a[o++] = 2; a[o++] = 2; a[o++] = 1; a[o++] = 1; a[o++] = 0; a[o++] = 0;
  a[o++] = 2;
  o += g_put_varint(a + o, J.nverts); o += g_put_varint(a + o, J.nf); a[o++] = (uint8_t)J.nad;
  o += g_put_varint(a + o, (uint32_t)J.nsym); o += g_put_varint(a + o, (uint32_t)J.nsplit);
  o += g_put_varint(a + o, (uint32_t)J.nev);
  { int last = 0;
    if (o + 10 * (uint32_t)J.nev + 64 > J.arena_cap) { J.status = -41; return; }
    for (int i = 0; i < J.nev; i++) { o += g_put_varint(a + o, (uint32_t)(J.ev_src[i] - last)); o += g_put_varint(a + o, (uint32_t)(J.ev_src[i] - J.ev_spl[i])); last = J.ev_src[i]; }
    if (J.nev > 0) { int nb = (J.nev + 7) / 8; for (int j = 0; j < nb; j++) { uint8_t v = 0; for (int k = 0; k < 8 && 8 * j + k < J.nev; k++) v |= (uint8_t)((J.ev_edge[8 * j + k] & 1) << k); a[o++] = v; } } }
  add_piece(J, a + b0, o - b0, total);
  add_piece(J, J.rb[0].buf + J.rb[0].off, J.rb[0].len, total);
  for (int i = 0; i < J.nad; i++) add_piece(J, J.rb[1 + i].buf + J.rb[1 + i].off, J.rb[1 + i].len, total);
  for (int i = 0; i < 6; i++) {
    b0 = o; o += g_put_varint(a + o, J.ctx_n[i]); add_piece(J, a + b0, o - b0, total);
    if (J.ctx_n[i] > 0) add_rans(J, i, total);
  }
  // attribute decoder headers
  b0 = o;
  const int dec_type[2] = { J.interior_seams[0] ? 1 : 0, J.interior_seams[1] ? 1 : 0 };
  a[o++] = (uint8_t)(1 + J.nad);
  a[o++] = 0xff; a[o++] = 0; a[o++] = 0;
  for (int i = 0; i < J.nad; i++) { a[o++] = (uint8_t)i; a[o++] = (uint8_t)dec_type[i]; a[o++] = 0; }
  a[o++] = 1; a[o++] = 0; a[o++] = 9; a[o++] = 3; a[o++] = 0; a[o++] = 0; a[o++] = 2;
  for (int i = 0; i < J.nad; i++) {
    a[o++] = 1;
    if (J.att_kind[i] == 0) { a[o++] = 3; a[o++] = 9; a[o++] = 2; a[o++] = 0; a[o++] = (uint8_t)(1 + i); a[o++] = 2; }
    else { a[o++] = 1; a[o++] = 9; a[o++] = 3; a[o++] = 0; a[o++] = (uint8_t)(1 + i); a[o++] = 3; }
  }
  // position values
  a[o++] = 1; a[o++] = 1; a[o++] = 1;
  add_piece(J, a + b0, o - b0, total);
  add_rans(J, 6, total);
  b0 = o;
  put_i32(a, o, J.wrap_lo[0]); put_i32(a, o, J.wrap_hi[0]);
  for (int k = 0; k < 3; k++) put_f32(a, o, g_float_unorder(J.pos_min_u[k]));
  put_f32(a, o, quant_range(J.pos_min_u, J.pos_max_u, 3)); a[o++] = (uint8_t)J.qp;
  for (int i = 0; i < J.nad; i++) {
    if (J.att_kind[i] == 0) {
      a[o++] = 5; a[o++] = 1; a[o++] = 1;
      add_piece(J, a + b0, o - b0, total);
      add_rans(J, 7, total);
      b0 = o; put_i32(a, o, (int32_t)J.n_ori); add_piece(J, a + b0, o - b0, total);
      add_piece(J, J.rb[3].buf + J.rb[3].off, J.rb[3].len, total);
      b0 = o;
      put_i32(a, o, J.wrap_lo[1]); put_i32(a, o, J.wrap_hi[1]);
      put_f32(a, o, g_float_unorder(J.uv_min_u[0])); put_f32(a, o, g_float_unorder(J.uv_min_u[1]));
      put_f32(a, o, quant_range(J.uv_min_u, J.uv_max_u, 2)); a[o++] = (uint8_t)J.qt;
    } else {
      const GOct ot = g_oct(J.qn);
      a[o++] = 6; a[o++] = 3; a[o++] = 1;
      add_piece(J, a + b0, o - b0, total);
      add_rans(J, 8, total);
      b0 = o; put_i32(a, o, ot.MAXQ); put_i32(a, o, ot.CEN); add_piece(J, a + b0, o - b0, total);
      add_piece(J, J.rb[4].buf + J.rb[4].off, J.rb[4].len, total);
      b0 = o; a[o++] = (uint8_t)J.qn;
    }
  }
  add_piece(J, a + b0, o - b0, total);
  J.out_len = total;
  if (total > J.out_cap) J.status = UVOL_E_NOSPACE;
}
// layout of a frame with sequential connectivity (see k_sq_*): header, index section, ONE attributes decoder
__global__ void __launch_bounds__(64) k_sq_layout(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.x];
  if (threadIdx.x != 0 || J.status != 0) return;
  uint8_t *a = J.arena; uint32_t o = 0, total = 0, b0 = 0;
  J.n_pieces = 0;
  a[o++] = 'D'; a[o++] = 'R'; a[o++] = 'A'; a[o++] = 'C'; a[o++] = 'O'; a[o++] = 2; a[o++] = 2; a[o++] = 1; a[o++] = 0; a[o++] = 0; a[o++] = 0;
  o += g_put_varint(a + o, J.nf_in); o += g_put_varint(a + o, J.sq_np); a[o++] = 1;                 // connectivity_method 1: indices stored directly
  add_piece(J, a + b0, o - b0, total);
  add_piece(J, J.sq_idx, J.sq_idx_bytes, total);
  b0 = o;
  a[o++] = 1;
  o += g_put_varint(a + o, (uint32_t)(1 + J.nad));
  a[o++] = 0; a[o++] = 9; a[o++] = 3; a[o++] = 0; a[o++] = 0;
  { int id = 1;
    if (J.has_uv) { a[o++] = 3; a[o++] = 9; a[o++] = 2; a[o++] = 0; a[o++] = (uint8_t)id++; }
    if (J.has_nrm) { a[o++] = 1; a[o++] = 9; a[o++] = 3; a[o++] = 0; a[o++] = (uint8_t)id++; } }
  a[o++] = 2; if (J.has_uv) a[o++] = 2; if (J.has_nrm) a[o++] = 3;
  a[o++] = 0; a[o++] = 1; a[o++] = 1;                                                                // position: DIFFERENCE, wrap, compressed
  add_piece(J, a + b0, o - b0, total);
  add_rans(J, 6, total);
  b0 = o; put_i32(a, o, J.wrap_lo[0]); put_i32(a, o, J.wrap_hi[0]);
  if (J.has_uv) {
    a[o++] = 0; a[o++] = 1; a[o++] = 1;
    add_piece(J, a + b0, o - b0, total);
    add_rans(J, 7, total);
    b0 = o; put_i32(a, o, J.wrap_lo[1]); put_i32(a, o, J.wrap_hi[1]);
  }
  if (J.has_nrm) {
    const GOct ot = g_oct(J.qn);
    a[o++] = 0; a[o++] = 3; a[o++] = 1;
    add_piece(J, a + b0, o - b0, total);
    add_rans(J, 8, total);
    b0 = o; put_i32(a, o, ot.MAXQ); put_i32(a, o, ot.CEN);
  }
  for (int k = 0; k < 3; k++) put_f32(a, o, g_float_unorder(J.pos_min_u[k]));
  put_f32(a, o, quant_range(J.pos_min_u, J.pos_max_u, 3)); a[o++] = (uint8_t)J.qp;
  if (J.has_uv) { put_f32(a, o, g_float_unorder(J.uv_min_u[0])); put_f32(a, o, g_float_unorder(J.uv_min_u[1])); put_f32(a, o, quant_range(J.uv_min_u, J.uv_max_u, 2)); a[o++] = (uint8_t)J.qt; }
  if (J.has_nrm) a[o++] = (uint8_t)J.qn;
  add_piece(J, a + b0, o - b0, total);
  J.out_len = total;
  if (total > J.out_cap) J.status = UVOL_E_NOSPACE;
}
// the frames' bitstreams are gathered back to back (16-byte aligned) so that the host fetches the whole batch with ONE copy
__global__ void __launch_bounds__(64) k_out_offsets(GeoJob *jobs, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t off = 0;
  for (int i = 0; i < n; i++) {
    jobs[i].out_pack_off = off;
    if (jobs[i].status != 0) continue;
    const uint64_t len = ((uint64_t)jobs[i].out_len + 15) & ~(uint64_t)15;
    if (off + len > jobs[i].slab_cap) { jobs[i].status = GEO_E_SLAB_FULL; continue; }      // the packed area is sized for typical streams
    off += len;
  }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_gather(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.z];
  if (J.status != 0) return;
  const uint32_t pc = blockIdx.y;
  if (pc >= J.n_pieces) return;
  const uint8_t *src = J.piece_ptr[pc]; uint8_t *dst = J.out_pack + J.out_pack_off + J.piece_off[pc]; const uint32_t len = J.piece_len[pc];
  for (uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x; i < len; i += gridDim.x * UVOL_BLOCK) dst[i] = src[i];
}

// ================================================================================================
// host side
// ================================================================================================
// workspace placement (see ws_collect / ws_place below)
struct WsItem { size_t slot, bytes; int first, last; size_t off; };
struct WsPlan {
  std::vector<uint64_t> key; std::vector<size_t> offs; size_t total = 0, zero = 0;
};
// placements by (bucketed) frame shape: the frames of a capture all differ a little in their counts (the reference's 250 frames have
// 26,144 - 27,979 vertices), and a first-fit placement per frame (~100 us, twice) would cost the host more than the GPU needs to encode
struct WsPlanCache { std::map<std::vector<uint64_t>, WsPlan> plans; };
// A lane = everything ONE group of frames needs while it is in flight: two streams (main + auxiliary), its events, its device
// buffers and the host-side record of the call it belongs to.  A call is cut into groups that run on different lanes, so that the
// bandwidth-bound front end of one group runs beside the latency-bound walkers of another (geo_encode_batch); lane 0 runs on the
// context's own stream.
struct GeoLane {
  hipStream_t stream = nullptr, aux = nullptr; bool own_stream = false;     // aux: valence replay runs beside renumber / seams / traversals
  hipEvent_t ev_walk = nullptr, ev_val = nullptr, ev_fe = nullptr;          // ev_fe: this group's front end (dedup + corner table) is done
  uvol_devbuf slab;       // all per-job workspaces
  uvol_devbuf inputs;     // staged inputs when the caller passes host pointers
  uvol_devbuf jobs;       // GeoJob[n]
  uvol_devbuf outs;       // output buffers
  std::vector<GeoJob> hjobs;
  uint8_t *pinned = nullptr; size_t pinned_cap = 0;
  uint32_t *counts = nullptr;          // device: {frames relabelled, frames with their predecessor's connectivity} of the group (k_relabel_decide)
  // the group in flight (submitted, not completed): its slice of the caller's arrays (the pointer arrays are copied: an enqueued call's arrays are gone by then)
  bool busy = false, on_device = false, full = false;
  std::vector<uvol_mesh> meshes; std::vector<uint8_t *> outp; std::vector<size_t> caps;
  size_t *out_lens = nullptr; int *status = nullptr; int n = 0, n_conc = 0;
  // GPU-resident form (uvol_encode_mesh_batch_dev_out): the packed output area is the CALLER's device buffer and the payload is not copied out
  uint8_t *ext_out = nullptr; size_t ext_cap = 0; size_t *ext_offs = nullptr; hipStream_t producer = nullptr; hipEvent_t ev_prod = nullptr;
  std::chrono::steady_clock::time_point t_enter; double t_prep = 0, t_enq = 0;
};
struct GeoState {
  std::vector<GeoLane *> lanes; int next_lane = 0;
  hipEvent_t fe_last = nullptr;        // front-end event of the group submitted last (the front ends of consecutive groups run one after the other)
  int deferred_rc = UVOL_OK;           // first error among groups completed on behalf of a later call (geo_flush returns it)
  WsPlanCache plan; std::vector<WsItem> items;     // workspace placements by frame shape
  size_t max_lds = 64 * 1024;
  int num_cu = 256;                    // CUs this context's streams may run on
};
static void geo_lane_free(GeoLane *L) {
  if (L->aux) { (void)hipStreamSynchronize(L->aux); (void)hipStreamDestroy(L->aux); }
  if (L->own_stream && L->stream) { (void)hipStreamSynchronize(L->stream); (void)hipStreamDestroy(L->stream); }
  for (uvol_devbuf *b : { &L->slab, &L->inputs, &L->jobs, &L->outs }) if (b->p) (void)hipFree(b->p);
  if (L->pinned) (void)hipHostFree(L->pinned);
  for (hipEvent_t e : { L->ev_walk, L->ev_val, L->ev_fe, L->ev_prod }) if (e) (void)hipEventDestroy(e);
  if (L->counts) (void)hipFree(L->counts);
  delete L;
}
// lane k of the context (created on first use; lane 0 = the context's stream)
static GeoLane *geo_lane(uvol_ctx *ctx, int k) {
  GeoState *G = ctx->geo;
  while ((int)G->lanes.size() <= k) {
    GeoLane *L = new GeoLane();
    if (G->lanes.empty()) L->stream = ctx->stream; else { if (uvol_make_stream(ctx, &L->stream) != hipSuccess) { delete L; return nullptr; } L->own_stream = true; }
    if (uvol_make_stream(ctx, &L->aux) != hipSuccess || hipEventCreate(&L->ev_walk) != hipSuccess || hipEventCreate(&L->ev_val) != hipSuccess ||
        hipEventCreateWithFlags(&L->ev_fe, hipEventDisableTiming) != hipSuccess) { geo_lane_free(L); return nullptr; }
    G->lanes.push_back(L);
  }
  return G->lanes[k];
}

// uvol_trim: the device workspaces of every lane go back to the device (they only grow: a lane that once ran a whole 2560-frame call keeps
// 130 GB); streams, events and the small job arrays stay.  The caller has completed the context's work (geo_flush).
int geo_trim(uvol_ctx *ctx) {
  GeoState *G = ctx->geo; if (!G) return UVOL_OK;
  for (GeoLane *L : G->lanes) {
    if (L->busy) continue;
    UVOL_HIP_CHECK(ctx, hipStreamSynchronize(L->stream)); UVOL_HIP_CHECK(ctx, hipStreamSynchronize(L->aux));
    for (uvol_devbuf *b : { &L->slab, &L->inputs, &L->outs }) if (b->p) { UVOL_HIP_CHECK(ctx, hipFree(b->p)); b->p = nullptr; b->cap = 0; }
  }
  return UVOL_OK;
}
int geo_create(uvol_ctx *ctx) {
  ctx->geo = new GeoState();
  if (!geo_lane(ctx, 0)) return UVOL_E_HIP;
#ifndef HIPEMU
  // the serial walkers keep their visited bitmaps in LDS: allow the full 160 KiB of a gfx950 CU
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) == hipSuccess && v > 0) ctx->geo->max_lds = (size_t)v;
  const size_t want = ctx->geo->max_lds;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && v > 0) ctx->geo->num_cu = v;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eb_walk<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eb_walk<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
  (void)hipGetLastError();
#else
  ctx->geo->max_lds = 160 * 1024;
#endif
  return UVOL_OK;
}
void geo_destroy(uvol_ctx *ctx) {
  if (!ctx->geo) return;
  GeoState *g = ctx->geo;
  for (GeoLane *L : g->lanes) geo_lane_free(L);
  delete g; ctx->geo = nullptr;
}

// entry capacity of a frame's per-vertex arrays = size of its vertex id space (ws_collect sizes them with it; geo_rec8 bounds the record fields with it)
static inline uint64_t geo_ecap(uint32_t n_pos, uint32_t n_uv, uint32_t n_nrm, uint32_t nf, bool full) {
  const uint64_t vmax = std::max<uint64_t>(n_pos, std::max<uint64_t>(n_uv, n_nrm)), nc = 3ull * nf;
  return full ? vmax + 2 * nc + 3 : std::min(vmax + 2 * nc + 3, vmax + vmax / 2 + 4096);
}
namespace {
inline uint32_t pow2_at_least(uint64_t v) { uint32_t c = 16; while (c < v) c <<= 1; return c; }

// ------------------------------------------------------------------------------------------------
// Workspace of one frame.  Every array has a lifetime [first, last] in pipeline phases; arrays whose lifetimes do not
// overlap share addresses (greedy first-fit over the live intervals, largest first), so a 200 k-face frame needs ~55 MB
// instead of 220 MB and a 288 GB GPU holds > 2000 frames in flight.  Arrays that must start out zero are pinned at the head
// of the workspace (one k_job_clear launch per batch) and are never shared.
// Per-vertex / per-entry arrays are sized for `ecap` entries (1.5 x the largest input attribute + slack) instead of the
// worst case 3 * faces; a mesh that needs more (non-manifold fans, every corner its own vertex) fails with GEO_E_WS_OVERFLOW
// on the device and is re-encoded alone with worst-case sizes (geo_encode_batch), so the compact layout never costs correctness.
// Phases (main stream order; the auxiliary stream runs events / valence replay / context scatter between PH_FTIME and PH_HIST):
enum { PH_DEDUP = 0, PH_FACES, PH_CT, PH_FANS0, PH_DENSE0, PH_WALK, PH_FTIME, PH_RENUM, PH_SEAMS, PH_DENSE1, PH_TRAV, PH_V2D, PH_QUANT,
       PH_PRED, PH_HIST, PH_ENT, PH_LAYOUT, PH_PINNED = -1 };

// Collects the arrays of job J (sizes from its input counts) and sets the capacities stored in J.  full = worst-case sizes.
// fmt0 / fmtT: record format of the walk table / of the three traversal tables (pack_face_records: 0, 1, 2)
void ws_collect(GeoJob &J, bool full, int fmt0, int fmtT, std::vector<WsItem> &items) {
  items.clear();
  const size_t nfi = J.nf_in, nc = 3 * nfi;
  const size_t vmax = std::max<size_t>(J.n_pos, std::max<size_t>(J.n_uv, J.n_nrm));
  // vertex ids are position ids + one id per further fan of a non-manifold position + (attribute tables) one per seam segment
  const size_t ecap = (size_t)geo_ecap(J.n_pos, J.n_uv, J.n_nrm, J.nf_in, full);
  J.ecap = (uint32_t)ecap;
  auto bitlen = [](uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; };
#define CARVE(field, T, count, first, last) items.push_back(WsItem{(size_t)((char *)&(field) - (char *)&J), (size_t)(count) * sizeof(T), (first), (last), 0})
  // dedup scratch, live in the first phase only.  Compact layout: the partitioned form (records by hash bin + the counts
  // matrix); worst-case layout (retries): the hash tables, zeroed right before use (k_dd_clear)
  for (int k = 0; k < 3; k++) {
    const uint32_t n = k == 0 ? J.n_pos : (k == 1 ? J.n_uv : J.n_nrm);
    J.dd_cap[k] = pow2_at_least(2ull * n + 2);
    J.dd_nblk[k] = (uint32_t)((n + DD_TILE - 1) / DD_TILE);
    J.dd_nb[k] = (uint32_t)std::min<uint64_t>(DD_MAXBINS, pow2_at_least(std::max<uint64_t>(1, n / 1024)));
    if (full) { CARVE(J.dd_tab[k], uint32_t, J.dd_cap[k], PH_DEDUP, PH_DEDUP); }
    else { CARVE(J.dd_part[k], uint4, (size_t)n + 1, PH_DEDUP, PH_DEDUP); CARVE(J.dd_cnt[k], uint32_t, (size_t)J.dd_nb[k] * J.dd_nblk[k] + 2, PH_DEDUP, PH_DEDUP); }
  }
  // locality relabelling (k_ms_*): keys, {key, index} records, counts matrices; new position ids / positions in that order; face maps
  if (J.relabel) {
    const size_t nmax = std::max<size_t>(J.n_pos, nfi);
    auto bits_of = [&](uint64_t v) { uint32_t b = 0; while (v) { b++; v >>= 1; } return b; };
    // bins of the first level: ~512 keys each, at most MS_MAXBINS (30-bit Morton keys / ids below n_pos: bin = the key's top bits)
    const uint32_t lb0 = std::min<uint32_t>(10, bits_of(J.n_pos / 512)), lb1 = std::min<uint32_t>(10, bits_of(nfi / 512));
    J.ms_sh[0] = 30 - lb0; J.ms_nb[0] = 1u << lb0; J.ms_nblk[0] = (uint32_t)((J.n_pos + MS_TILE - 1) / MS_TILE);
    { const uint32_t kb = bits_of(J.n_pos ? J.n_pos - 1 : 0); J.ms_sh[1] = kb > lb1 ? kb - lb1 : 0; }
    J.ms_nb[1] = 0; J.ms_nblk[1] = 0;                      // set by k_relabel_decide
    const uint32_t ms_nblk1 = (uint32_t)((nfi + MS_TILE - 1) / MS_TILE);
    CARVE(J.ms_key[0], uint32_t, (size_t)J.n_pos + 1, PH_DEDUP, PH_DEDUP); CARVE(J.ms_key[1], uint32_t, nfi + 1, PH_FACES, PH_FACES);
    CARVE(J.ms_part, uint2, nmax + 1, PH_DEDUP, PH_FACES);
    CARVE(J.ms_cnt, uint32_t, (size_t)MS_MAXBINS * std::max(J.ms_nblk[0], ms_nblk1) + 2, PH_DEDUP, PH_FACES);
    CARVE(J.prank, uint32_t, (size_t)J.n_pos + 1, PH_DEDUP, PH_FACES); CARVE(J.pos_s, float, 3 * (size_t)J.n_pos + 3, PH_DEDUP, PH_QUANT);
    CARVE(J.fperm, uint32_t, nfi + 1, PH_FACES, PH_FACES); CARVE(J.cidx, uint32_t, nfi + 1, PH_FACES, PH_FACES);
    CARVE(J.forig, int32_t, nfi + 1, PH_FACES, PH_CT); CARVE(J.s_of_o, int32_t, nfi + 1, PH_FACES, PH_WALK);
  }
  if (J.seq) {                                              // sequential connectivity: hash table, per-corner maps, the index section
    J.sq_cap = pow2_at_least(2ull * nc + 2);
    CARVE(J.sq_keys, unsigned long long, J.sq_cap, PH_DEDUP, PH_LAYOUT); CARVE(J.sq_val, uint32_t, J.sq_cap, PH_DEDUP, PH_LAYOUT);
    CARVE(J.sq_pu, int32_t, nc + 3, PH_DEDUP, PH_LAYOUT); CARVE(J.sq_first, int32_t, nc + 3, PH_DEDUP, PH_LAYOUT); CARVE(J.sq_pid, int32_t, nc + 3, PH_DEDUP, PH_LAYOUT);
    CARVE(J.sq_cop, int32_t, ecap + 1, PH_DEDUP, PH_LAYOUT); CARVE(J.sq_flag, uint8_t, nc + 3, PH_DEDUP, PH_LAYOUT); CARVE(J.sq_idx, uint8_t, 4 * nc + 16, PH_DEDUP, PH_LAYOUT);
  }
  // ---- pinned, zero-initialised ----
  CARVE(J.he_start, uint32_t, (size_t)J.n_pos + 1, PH_PINNED, PH_PINNED);
  CARVE(J.vvis, uint8_t, ecap / 8 + 64, PH_PINNED, PH_PINNED);
  for (int i = 0; i < 2; i++) CARVE(J.vseam[i], uint32_t, ecap / 32 + 2, PH_PINNED, PH_PINNED);      // one bit per vertex: the whole map stays in L2
  for (int t = 0; t < 3; t++) CARVE(J.t_vvis[t], uint8_t, ecap / 8 + 64, PH_PINNED, PH_PINNED);
  CARVE(J.fvis, uint8_t, nfi / 8 + 64, PH_PINNED, PH_PINNED);                                           // face-visited bits of the lane-per-walker kernels on per-face records
  for (int t = 0; t < 3; t++) CARVE(J.t_fvis[t], uint8_t, nfi / 8 + 64, PH_PINNED, PH_PINNED);
  for (int s = 0; s < GEO_NSTREAM; s++) {
    const int q = s == 6 ? J.qp : (s == 7 ? J.qt : J.qn);
    J.rs[s].alpha_cap = s < 6 ? 8 : (1u << (q + 1)) + 8;
    CARVE(J.rs[s].freq, uint32_t, J.rs[s].alpha_cap, PH_PINNED, PH_PINNED);
  }
  // ---- scan scratch (tiny, kept for the whole batch) ----
  CARVE(J.bsum, uint32_t, nc / UVOL_BLOCK + 8, PH_DEDUP, PH_LAYOUT); CARVE(J.bsum2, uint32_t, nc / UVOL_BLOCK + 8, PH_DEDUP, PH_LAYOUT);
  // ---- K2 / K3 ----
  { const int cl = J.seq ? PH_LAYOUT : PH_FACES;          // the sequential path reads the canonical ids when it quantises the points
    CARVE(J.canon[0], uint32_t, J.n_pos + 1, PH_DEDUP, cl); CARVE(J.canon[1], uint32_t, J.n_uv + 1, PH_DEDUP, cl); CARVE(J.canon[2], uint32_t, J.n_nrm + 1, PH_DEDUP, cl); }
  CARVE(J.keep, uint8_t, nfi + 1, PH_FACES, PH_FACES);
  CARVE(J.cp, int32_t, nc + 3, PH_FACES, PH_RENUM); CARVE(J.cu, int32_t, nc + 3, PH_FACES, PH_RENUM); CARVE(J.cn, int32_t, nc + 3, PH_FACES, PH_RENUM);
  CARVE(J.he_cur, uint32_t, (size_t)J.n_pos + 1, PH_CT, PH_FANS0); CARVE(J.he_ent, unsigned long long, nc + 1, PH_CT, PH_FANS0);      // k_vert0 walks the buckets
  { // partitioned bucket build (compact layout, ranges of <= HE_MAXVPB vertices): records + counts matrix, live in PH_CT only
    uint32_t vpb = 512; while ((uint64_t)vpb * HE_MAXBINS < (uint64_t)J.n_pos) vpb *= 2;
    J.he_vpb = (!full && vpb <= HE_MAXVPB && J.n_pos > 0) ? vpb : 0u;
    J.he_nb = J.he_vpb ? (J.n_pos + vpb - 1) / vpb : 0u; J.he_nblk = J.he_vpb ? (uint32_t)((nc + HE_TILE - 1) / HE_TILE) : 0u;
    if (J.he_vpb) { CARVE(J.he_part, uint32_t, 3 * nc + 4, PH_CT, PH_CT); CARVE(J.he_cnt, uint32_t, (size_t)J.he_nb * J.he_nblk + 2, PH_CT, PH_CT); }
  }
  const int aux_last = J.late_join ? PH_PRED : PH_SEAMS;            // what the auxiliary stream (valence replay, context scatter) reads lives until its join
  CARVE(J.opp, int32_t, nc + 3, PH_CT, aux_last);
  CARVE(J.vert, int32_t, nc + 3, PH_FANS0, PH_SEAMS);
  // ---- K4 ----
  auto rec_size = [&](int fmt) { return (size_t)(fmt == 2 ? 16 : (fmt == 1 ? 32 : 64)) * (nfi + 1); };
  const size_t rec_bytes = rec_size(fmtT);
  CARVE(J.rec[0], uint8_t, rec_size(fmt0), PH_DENSE0, PH_WALK); CARVE(J.vopen_d[0], uint8_t, ecap, PH_FANS0, PH_DENSE1);
  for (int w = 1; w < 4; w++) CARVE(J.rec[w], uint8_t, rec_bytes, PH_DENSE1, PH_V2D);
  CARVE(J.ring_d, int32_t, ecap, PH_FANS0, PH_SEAMS);
  CARVE(J.face_time, int32_t, nfi + 1, PH_DENSE0, aux_last);
  CARVE(J.proc, int32_t, nfi + 1, PH_WALK, aux_last); CARVE(J.symb, uint8_t, nfi + 64, PH_WALK, aux_last);
  CARVE(J.initc, int32_t, nfi + 1, PH_WALK, PH_RENUM); CARVE(J.stack, int32_t, nfi + 2, PH_WALK, PH_WALK); CARVE(J.start_bits, uint8_t, nfi + 1, PH_WALK, PH_ENT);
  // auxiliary stream (forked after PH_FTIME, joined before PH_HIST)
  CARVE(J.evcnt, uint8_t, nfi + 1, PH_RENUM, PH_SEAMS);
  CARVE(J.ev_src, int32_t, 2 * nfi + 2, PH_RENUM, PH_LAYOUT); CARVE(J.ev_spl, int32_t, 2 * nfi + 2, PH_RENUM, PH_LAYOUT); CARVE(J.ev_edge, uint8_t, 2 * nfi + 2, PH_RENUM, PH_LAYOUT);
  CARVE(J.vval, int32_t, ecap + nfi + 3, PH_RENUM, aux_last); CARVE(J.c2vm, int32_t, nc + 3, PH_RENUM, aux_last); CARVE(J.ctx_of, uint8_t, nfi + 64, PH_RENUM, aux_last);
  for (int i = 0; i < 6; i++) CARVE(J.ctx_sym[i], uint32_t, nfi + 1, PH_RENUM, PH_ENT);
  // ---- renumbering, seams ----
  CARVE(J.new_of_old, int32_t, nc + 3, PH_RENUM, PH_RENUM); CARVE(J.nopp, int32_t, nc + 3, PH_RENUM, PH_PRED);
  CARVE(J.npid, int32_t, nc + 3, PH_RENUM, PH_QUANT); CARVE(J.nuid, int32_t, nc + 3, PH_RENUM, PH_QUANT); CARVE(J.nnid, int32_t, nc + 3, PH_RENUM, PH_QUANT);
  CARVE(J.bvert, int32_t, nc + 3, PH_RENUM, PH_PRED);
  for (int i = 0; i < 2; i++) {
    CARVE(J.seam[i], uint8_t, nc + 3, PH_RENUM, PH_PRED); CARVE(J.seam_bits[i], uint8_t, nc + 3, PH_SEAMS, PH_ENT);
    CARVE(J.avert[i], int32_t, nc + 3, PH_SEAMS, PH_PRED);
  }
  CARVE(J.elig, uint8_t, nc + 3, PH_RENUM, PH_SEAMS);
  // ---- K5, K1, K6 ----
  for (int t = 0; t < 3; t++) { CARVE(J.order[t], int32_t, ecap, PH_TRAV, PH_PRED); CARVE(J.v2d[t], int32_t, ecap, PH_V2D, PH_PRED); CARVE(J.t_stack[t], int32_t, nfi + 2, PH_TRAV, PH_TRAV); }
  CARVE(J.P, int32_t, 3 * ecap, PH_QUANT, PH_PRED); CARVE(J.U, int32_t, 2 * ecap, PH_QUANT, PH_PRED); CARVE(J.O, int32_t, 2 * ecap, PH_QUANT, PH_PRED);
  CARVE(J.fnorm, int32_t, J.n_nrm ? (J.qp <= 15 ? 3 : 6) * nfi + 6 : 2, PH_PRED, PH_PRED);
  CARVE(J.sym_pos, uint32_t, 3 * ecap, PH_PRED, PH_ENT); CARVE(J.sym_uv, uint32_t, 2 * ecap, PH_PRED, PH_ENT); CARVE(J.sym_nrm, uint32_t, 2 * ecap, PH_PRED, PH_ENT);
  CARVE(J.has_ori, uint8_t, ecap, PH_PRED, PH_PRED); CARVE(J.ori_val, uint8_t, ecap, PH_PRED, PH_PRED); CARVE(J.ori_c, uint8_t, ecap, PH_PRED, PH_PRED);
  CARVE(J.ori_bits, uint8_t, ecap, PH_PRED, PH_ENT); CARVE(J.flips, uint8_t, ecap, PH_PRED, PH_ENT);
  // ---- K7 ----
  for (int s = 0; s < GEO_NSTREAM; s++) {
    RansStream &S = J.rs[s];
    const size_t nsym = s < 6 ? nfi : (s == 6 ? 3 * ecap : 2 * ecap);
    CARVE(S.probs, uint32_t, S.alpha_cap, PH_HIST, PH_LAYOUT); CARVE(S.cum, uint32_t, S.alpha_cap, PH_HIST, PH_LAYOUT);
    CARVE(S.head, uint8_t, 3 * (size_t)S.alpha_cap + 32, PH_HIST, PH_LAYOUT);
    CARVE(S.tab, uint4, S.alpha_cap, PH_HIST, PH_ENT);
    S.pay_cap = (uint32_t)(3 * nsym + 256); CARVE(S.pay, uint8_t, S.pay_cap, PH_ENT, PH_LAYOUT);
    // counting-sort scratch of k_rans_tables: (precision + 2) counters + one slot per symbol of the alphabet
    const int bl = bitlen(S.alpha_cap), pb = std::min(20, std::max(12, 3 * bl / 2));
    CARVE(S.scratch, uint32_t, ((size_t)1 << pb) + 4 + S.alpha_cap, PH_HIST, PH_HIST);
    S.syms = nullptr; S.n = 0; S.max_sym = 0; S.head_len = 0; S.pay_len = 0; S.pay_off = 0; S.prec_bits = 12;
  }
  for (int b = 0; b < GEO_NRABS; b++) {
    RabsStream &B = J.rb[b];
    const size_t nb = b == 0 ? nfi : (b < 3 ? nc : ecap);
    B.cap = (uint32_t)(nb / 4 + nb / 8 + 256); CARVE(B.buf, uint8_t, B.cap, PH_ENT, PH_LAYOUT);
    B.bits = nullptr; B.n = 0; B.zeros = 0; B.off = 0; B.len = 0;
  }
  J.arena_cap = (uint32_t)(20 * nfi + 1024); CARVE(J.arena, uint8_t, J.arena_cap, PH_LAYOUT, PH_LAYOUT);
#undef CARVE
}

// first-fit placement over the lifetime intervals; pinned items first (they form the zeroed head)
void ws_place(std::vector<WsItem> &items, WsPlan &P) {
  std::vector<UvolWsItem> w(items.size());
  for (size_t i = 0; i < items.size(); i++) w[i] = UvolWsItem{ items[i].bytes, items[i].first, items[i].last, 0 };
  P.total = uvol_ws_place(w, &P.zero, PH_LAYOUT + 1, "geometry encode");
  P.offs.resize(items.size());
  for (size_t i = 0; i < items.size(); i++) { items[i].off = w[i].off; P.offs[i] = w[i].off; }
}

// Lays out one job's workspace (sizes + capacities always; pointers when base != nullptr).  The capacities stored in J come from its
// own counts; the PLACEMENT is the one of the frame's shape bucket - its counts rounded up to multiples of 1024 (2048 faces), at most
// 1 % more bytes at 100 k vertices - so that the differing frames of a sequence share a handful of cached placements.
const WsPlan &layout_job(GeoJob &J, uint8_t *base, bool full, int fmt0, int fmtT, WsPlanCache &C, std::vector<WsItem> &items) {
  ws_collect(J, full, fmt0, fmtT, items);
  auto up = [](uint32_t v, uint32_t q) { return (uint64_t)((v + (uint64_t)q - 1) / q) * q; };
  const uint64_t flags = (uint64_t)J.qp | ((uint64_t)J.qt << 8) | ((uint64_t)J.qn << 16) | ((uint64_t)full << 24) | ((uint64_t)(J.relabel != 0) << 26) |
                         ((uint64_t)(J.seq != 0) << 27) | ((uint64_t)(J.late_join != 0) << 28) | ((uint64_t)fmt0 << 29) | ((uint64_t)fmtT << 31);      // everything ws_collect's sizes AND lifetimes depend on
  std::vector<uint64_t> key = { up(J.nf_in, 2048), up(J.n_pos, 1024), up(J.n_uv, 1024), up(J.n_nrm, 1024), flags, items.size(), 0 };
  auto it = C.plans.find(key);
  if (it == C.plans.end()) {
    GeoJob R = J; R.nf_in = (uint32_t)std::min<uint64_t>(key[0], 1u << 26); R.n_pos = (uint32_t)key[1]; R.n_uv = (uint32_t)key[2]; R.n_nrm = (uint32_t)key[3];
    std::vector<WsItem> ri; ws_collect(R, full, fmt0, fmtT, ri);
    bool ok = ri.size() == items.size();
    for (size_t i = 0; ok && i < ri.size(); i++) ok = ri[i].slot == items[i].slot && ri[i].bytes >= items[i].bytes && ri[i].first == items[i].first && ri[i].last == items[i].last;
    if (!ok) {                                             // a count on a structural boundary (an array the rounded shape does not have): exact placement for this shape
      key = { J.nf_in, J.n_pos, J.n_uv, J.n_nrm, flags, items.size(), 1 };
      it = C.plans.find(key);
      ri = items;
    }
    if (it == C.plans.end()) {
      if (C.plans.size() >= 512) C.plans.clear();
      WsPlan P; ws_place(ri, P); P.key = key;
      it = C.plans.emplace(key, std::move(P)).first;
    }
  }
  const WsPlan &P = it->second;
  if (base) for (size_t i = 0; i < items.size(); i++) *reinterpret_cast<uint8_t **>((char *)&J + items[i].slot) = base + P.offs[i];
  return P;
}
}  // namespace

// Upper bound of a .drc for quantisation bits <= 16: per value <= 2.5 bytes of rANS payload (20-bit precision) + 3 bytes of
// probability table per distinct symbol, 21 values per face at most; connectivity symbols, seam / orientation bits and the
// split events stay below 20 bytes per face; fixed headers and the zero runs of three 2^17-symbol tables below 64 KiB.
size_t uvol_mesh_bound(const uvol_mesh *m) {
  if (!m) return 0;
  return 65536 + (size_t)m->n_faces * 144;
}

#define LAUNCH(k, grid, block, ...)                                                              \
  do {                                                                                           \
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch %s\n", #k); fflush(stderr); }              \
    hipLaunchKernelGGL(k, grid, block, 0, ctx->stream, __VA_ARGS__);                             \
    if (uvol_debug()) { hipError_t e_ = hipStreamSynchronize(ctx->stream); if (e_ != hipSuccess) { fprintf(stderr, "[uvol] %s FAILED: %s\n", #k, hipGetErrorString(e_)); fflush(stderr); } \
      GeoJob dbg_; (void)hipMemcpy(&dbg_, dj, sizeof(GeoJob), hipMemcpyDeviceToHost); \
      fprintf(stderr, "[uvol]   job0 status %d nf %u nverts %u ne %u %u %u ne_uv %u has_ori %p bsum %p elig %p n_ori %u\n", dbg_.status, dbg_.nf, dbg_.nverts, dbg_.ne[0], dbg_.ne[1], dbg_.ne[2], dbg_.ne_uv, (void*)dbg_.has_ori, (void*)dbg_.bsum, (void*)dbg_.elig, dbg_.n_ori); fflush(stderr); } \
  } while (0)

#define LAUNCH_ON(stream_, k, grid, block, ...)                                                  \
  do {                                                                                           \
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch %s (aux)\n", #k); fflush(stderr); }        \
    hipLaunchKernelGGL(k, grid, block, 0, stream_, __VA_ARGS__);                                 \
    if (uvol_debug()) { hipError_t e_ = hipStreamSynchronize(stream_); if (e_ != hipSuccess) { fprintf(stderr, "[uvol] %s FAILED: %s\n", #k, hipGetErrorString(e_)); fflush(stderr); } } \
  } while (0)
#define LAUNCH_SM(k, grid, block, shmem, ...)                                                    \
  do {                                                                                           \
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch %s (lds %zu)\n", #k, (size_t)(shmem)); fflush(stderr); } \
    hipLaunchKernelGGL(k, grid, block, shmem, ctx->stream, __VA_ARGS__);                         \
    if (uvol_debug()) { hipError_t e_ = hipStreamSynchronize(ctx->stream); if (e_ != hipSuccess) { fprintf(stderr, "[uvol] %s FAILED: %s\n", #k, hipGetErrorString(e_)); fflush(stderr); } } \
  } while (0)

// zero the dedup hash tables / half-edge counts / visited maps / histograms at the head of every job's workspace
__global__ void __launch_bounds__(UVOL_BLOCK) k_job_clear(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  uint4 *p = reinterpret_cast<uint4 *>(J.ws_base);
  const size_t n16 = (size_t)(J.ws_zero / 16);
  for (size_t i = (size_t)blockIdx.x * UVOL_BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * UVOL_BLOCK) p[i] = make_uint4(0, 0, 0, 0);
}

// zero the three dedup hash tables of the jobs of one group (grid y = job, z = table)
__global__ void __launch_bounds__(UVOL_BLOCK) k_dd_clear(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  uint4 *p = reinterpret_cast<uint4 *>(J.dd_tab[blockIdx.z]);
  const size_t n16 = (size_t)J.dd_cap[blockIdx.z] / 4;                   // capacities are powers of two >= 4, tables 256-byte aligned
  for (size_t i = (size_t)blockIdx.x * UVOL_BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * UVOL_BLOCK) p[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && threadIdx.x == 0) for (uint32_t i = (uint32_t)n16 * 4; i < J.dd_cap[blockIdx.z]; i++) J.dd_tab[blockIdx.z][i] = 0;
}

// How the serial walkers of a batch run.  Wave-per-walker (one lane of a wave per walker, visited bitmaps in LDS) is the
// faster form per walker (one dependent load per face, 0.35 - 0.5 us) but a CU's LDS holds 3 of them; lane-per-walker (SIMT,
// nothing in LDS, 0.55 - 0.65 us per face at 16 lanes per wave) has no such cap.  So: LDS walkers while all of a launch's
// walkers are resident at once, SIMT walkers beyond that (measured on 720 frames in one launch: traversals 265 -> 135 ms).
// UVOL_SIMT_W=<1..64> forces the SIMT form with that many lanes per wave, UVOL_WALK_FORCE=global its one-lane-per-wave form
// (what a mesh too large for LDS gets), UVOL_WALK_FORCE=vglobal LDS walkers with their vertex bitmap in global memory (tests).
struct WalkPlan { int simt_w; size_t lds; int vcw; };
// locality relabelling: 2 = per frame, decided on the device (k_coherence: frames stored coherently skip it); UVOL_RELABEL=1 / 0 (tests, diagnostic) force it on / off
static inline int geo_relabel_mode() { static const int v = [] { const char *e = getenv("UVOL_RELABEL"); return !e ? 2 : (*e == '0' ? 0 : (*e == '1' ? 1 : 2)); }(); return v; }
static inline bool geo_relabel_on() { return geo_relabel_mode() != 0; }
static inline int geo_walk_pf() { static const int v = [] { const char *e = getenv("UVOL_WALK_PF"); return (e && *e == '0') ? 0 : 1; }(); return v; }     // UVOL_WALK_PF=0 (diagnostic): LDS walkers without their prefetch wave
static inline int geo_simt_env() { static const int w = [] { const char *e = getenv("UVOL_SIMT_W"); int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }(); return w; }
static WalkPlan walk_plan(const GeoState *G, uint32_t max_nfi, uint32_t max_vals, size_t n_walkers, bool vertex_bits_global = false) {
  WalkPlan P{0, 0, 0};
  const size_t walk_fw = ((size_t)max_nfi + 31) / 32;
  // Vertex bitmap capacity.  At least the largest attribute array of the batch + 6 % (vertices split at seams and
  // non-manifold fans; a table that still exceeds it keeps its vertex bitmap in global memory); then rounded UP to
  // whatever fits the same number of walkers per CU, so the slack of the LDS slot is not wasted.
  const size_t lds_cu = 150 * 1024 /* what several workgroups can share of a CU's 160 KiB (measured: 3 x 53 KiB does not fit) */, fw_bytes = walk_fw * 4 + (WALK_STG_DWORDS + WALK_PUB_DWORDS) * 4 + 16 /* + output staging, position word */;
  const size_t v_min_bytes = std::min<size_t>((((size_t)max_vals + max_vals / 16 + 31) / 32 + 2) * 4, ((3 * (size_t)max_nfi + 31) / 32) * 4);
  static const int walk_force = [] { const char *e = getenv("UVOL_WALK_FORCE"); return !e ? 0 : (!strcmp(e, "vglobal") ? 1 : (!strcmp(e, "global") ? 2 : 0)); }();
  const bool vglobal = walk_force == 1 || vertex_bits_global;
  size_t per_cu = lds_cu / (fw_bytes + (vglobal ? 4 : v_min_bytes)); if (per_cu < 1) per_cu = 1;
  const size_t slot = (lds_cu / per_cu) & ~(size_t)1023;
  const size_t walk_vcw = vglobal ? 1 : (slot > fw_bytes + v_min_bytes ? (slot - fw_bytes) / 4 : v_min_bytes / 4);
  P.lds = (((walk_fw + walk_vcw + 3) & ~(size_t)3) + WALK_STG_DWORDS + WALK_PUB_DWORDS) * 4; P.vcw = (int)walk_vcw;
  const bool lds_fits = P.lds <= G->max_lds;
  if (geo_simt_env() > 0) P.simt_w = geo_simt_env();
  else if (walk_force == 2 || !lds_fits) P.simt_w = 1;
  else if (walk_force == 0 && n_walkers > per_cu * (size_t)G->num_cu) P.simt_w = n_walkers >= 1024 ? 16 : 4;
  return P;
}
// 8-byte corner records (RecOps<true>): every corner code (< 4 * faces, signed field: 20 bits + sign) and every
// vertex id << 1 | open (unsigned 21-bit field) of the batch must fit: faces < 2^18 AND the vertex id space below 2^20.  Ids are
// position-based (n_pos + extra fans + seam segments, bounded by the workspace's entry capacity `ecap`), NOT bounded by the
// face count: a 2000-face mesh that references position 2^20 + 5 needs the 16-byte format.  UVOL_REC16=1 (tests) forces it.
static inline bool geo_rec8(uint32_t max_nfi, uint64_t max_ids) {
  static const bool force16 = [] { const char *e = getenv("UVOL_REC16"); return e && *e == '1'; }();
  return !force16 && 4ull * max_nfi < (1ull << 20) && max_ids < (1ull << 20);
}
// the decode path sizes its record tables with the same rule; its vertex ids are dense (< 3 * faces)
bool geo_records8(uint32_t max_nfi) { return geo_rec8(max_nfi, 3ull * max_nfi); }
// UVOL_FACE_BITS=1 (diagnostic, tests): face-visited bits in an array of their own instead of bit 63 of the per-face record.  Measured and
// NOT the default: traversals 275 against 251 ms, geometry alone 3074 against 3169 frames/s, full path 2580 against 2745 - the two extra
// 4-byte loads per step cost more than the clean record lines save (tools/experiments/exp_r4h.sh).
static inline bool geo_face_bits() { static const bool v = [] { const char *e = getenv("UVOL_FACE_BITS"); return e && *e == '1'; }(); return v; }
static inline bool geo_rec_face_off() { static const bool v = [] { const char *e = getenv("UVOL_REC_FACE"); return e && *e == '0'; }(); return v; }      // UVOL_REC_FACE=0 (diagnostic, tests): corner records in the lane-per-walker kernels too
static inline bool geo_walk_ld() { static const bool v = [] { const char *e = getenv("UVOL_WALK_LD"); return e && *e == '1'; }(); return v; }
static void launch_traversals(uvol_ctx *ctx, GeoJob *dj, int n, const WalkPlan &P, int r8) {
  const unsigned N = (unsigned)n;
  if (P.simt_w) {
    const unsigned W = (unsigned)P.simt_w, nb = (3 * N + W - 1) / W;
    if (r8 == 2) { if (geo_face_bits()) LAUNCH((k_traverse_simt_f16<true, 0>), dim3(nb), dim3(64), dj, n, (int)W); else if (geo_walk_ld()) LAUNCH((k_traverse_simt_f16<false, 1>), dim3(nb), dim3(64), dj, n, (int)W); else LAUNCH((k_traverse_simt_f16<false, 0>), dim3(nb), dim3(64), dj, n, (int)W); }
    else if (r8) LAUNCH((k_traverse_simt<true>), dim3(nb), dim3(64), dj, n, (int)W); else LAUNCH((k_traverse_simt<false>), dim3(nb), dim3(64), dj, n, (int)W);
  }
  else if (r8) LAUNCH_SM((k_traverse<true>), dim3(3, N), dim3(128), P.lds, dj, P.vcw, geo_walk_pf() ? 2 : 0);
  else LAUNCH_SM((k_traverse<false>), dim3(3, N), dim3(128), P.lds, dj, P.vcw, geo_walk_pf() ? 2 : 0);
}
// attribute sequencing of a prepared GeoJob array (tables 1..3): corner records, DepthFirstTraverser, inverse maps.
// Shared with the decode path (geom_decode.hip), which fills the same job fields from a decoded corner table.
int geo_run_traversals(uvol_ctx *ctx, GeoJob *dj, int n, uint32_t max_nfi, uint32_t max_vals) {
  GeoState *G = ctx->geo;
  const unsigned N = (unsigned)n, bf = uvol_blocks(max_nfi), bc = uvol_blocks((size_t)3 * max_nfi);
  WalkPlan P = walk_plan(G, max_nfi, max_vals, (size_t)3 * N);
  // files of a batch are unrelated meshes as far as the decoder knows: one traverser per wave (several per wave pay for each other's
  // rare paths and misses: 953 against 293 ms per 2560 distinct frames with 16 / 1), and one 16-byte record per face where the fields allow it
  if (P.simt_w > 1 && geo_simt_env() == 0) P.simt_w = 1;
  const int r8 = geo_records8(max_nfi) ? ((P.simt_w && !geo_rec_face_off()) ? 2 : 1) : 0;
  for (int w = 1; w <= 3; w++) LAUNCH(k_pack_faces, dim3(bf, N), dim3(UVOL_BLOCK), dj, w, r8);
  launch_traversals(ctx, dj, n, P, r8);
  LAUNCH(k_v2d, dim3(bc, N, 3), dim3(UVOL_BLOCK), dj, r8);
  return UVOL_OK;
}

// device workspace one frame of these dimensions holds while it is in flight (compact layout + its share of the packed output area)
extern "C" size_t uvol_mesh_workspace(const uvol_ctx *ctx, const uvol_mesh *m) {
  if (!ctx || !m || !m->n_faces) return 0;
  GeoJob J{}; J.relabel = geo_relabel_mode(); J.n_pos = m->n_pos; J.nf_in = m->n_faces; J.n_uv = (m->uv && m->idx_uv) ? m->n_uv : 0; J.n_nrm = (m->nrm && m->idx_nrm) ? m->n_nrm : 0;
  J.qp = ctx->prm.q_position_attr; J.qt = ctx->prm.q_texture_attr; J.qn = ctx->prm.q_normal_attr;
  WsPlanCache P; std::vector<WsItem> items;
  const int fmt = geo_rec8(m->n_faces, geo_ecap(J.n_pos, J.n_uv, J.n_nrm, J.nf_in, false)) ? 2 : 0;       // what a frame of a large batch holds (lane-per-walker kernels, one record per face)
  return layout_job(J, nullptr, false, fmt, fmt, P, items).total + 32768 + 8 * (size_t)m->n_faces + sizeof(GeoJob);
}
// stages of a batch with sequential connectivity, between k_minmax and the layout (all parallel; see k_sq_*)
static int geo_encode_sequential(uvol_ctx *ctx, GeoJob *dj, int n, bool full, uint32_t max_nfi, uint32_t max_vals, uint32_t max_ecap, uint64_t algo_in) {
  const unsigned N = (unsigned)n, bc = uvol_blocks((size_t)3 * max_nfi), bv = uvol_blocks(max_vals), be = uvol_blocks(std::min<size_t>(max_ecap, (size_t)3 * max_nfi));
  const uvol_params &prm = ctx->prm;
  {
    uvol_ctx::Scope sc(ctx, "geo.k2_dedup", algo_in);
    if (!full) {
      uint32_t slots = DD_SLOTS; { const char *e = getenv("UVOL_DD_SLOTS"); const int v = e ? atoi(e) : 0; if (v >= 4 && v <= DD_SLOTS && !(v & (v - 1))) slots = (uint32_t)v; }
      const unsigned bt = (unsigned)((max_vals + DD_TILE - 1) / DD_TILE), nbm = (unsigned)std::min<uint64_t>(DD_MAXBINS, pow2_at_least(std::max<uint64_t>(1, max_vals / 1024)));
      LAUNCH(k_dd_count, dim3(bt, N, 3), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_dd_scan, dim3(1, N, 3), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_dd_scatter, dim3(bt, N, 3), dim3(UVOL_BLOCK), dj);
      if (max_vals / std::max(1u, nbm) <= 1100u && slots >= 2048u) LAUNCH((k_dd_resolve<2048, 4>), dim3(nbm, N, 3), dim3(UVOL_BLOCK), dj, 2048u);
      else LAUNCH((k_dd_resolve<DD_SLOTS, 6>), dim3(nbm, N, 3), dim3(UVOL_BLOCK), dj, slots);
    } else {
      LAUNCH(k_dd_clear, dim3(16, N, 3), dim3(UVOL_BLOCK), dj);
      for (int ph = 0; ph < 2; ph++) { LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 0, ph); LAUNCH(k_dedup<2>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 1, ph); LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 2, ph); }
    }
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.seq_points", (uint64_t)3 * max_nfi * 12);
    for (int level = 0; level < 2; level++) {
      LAUNCH(k_sq_clear, dim3(256, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_sq_hash, dim3(bc, N), dim3(UVOL_BLOCK), dj, level, 0);
      LAUNCH(k_sq_hash, dim3(bc, N), dim3(UVOL_BLOCK), dj, level, 1);
    }
    for (int step = 0; step < 4; step++) {
      LAUNCH(k_sq_points, dim3(bc, N), dim3(UVOL_BLOCK), dj, step);
      if (step == 0 || step == 2) LAUNCH(k_scan_sums, dim3(1, N), dim3(UVOL_BLOCK), dj, (int)SCAN_SEQ);
    }
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.k1_quantize", algo_in);
    LAUNCH(k_sq_quant, dim3(be, N, 3), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_sq_pred, dim3(be, N, 3), dim3(UVOL_BLOCK), dj);
  }
  {
    uvol_ctx::Scope sc0(ctx, "geo.k7_hist_tables", 0);
    LAUNCH(k_hist, dim3(uvol_blocks((size_t)9 * max_nfi, 16 * UVOL_BLOCK), GEO_NSTREAM, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_rans_tables, dim3(GEO_NSTREAM, N), dim3(64), dj);
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.k7_entropy_encode", 0);
    unsigned W = 4; while (W < 64 && 5u * N > 512u * W) W *= 2;
    LAUNCH(k_rans_recip, dim3(uvol_blocks(((size_t)2 << std::max(prm.q_position_attr, std::max(prm.q_texture_attr, prm.q_normal_attr))) + 8), GEO_NSTREAM, N), dim3(UVOL_BLOCK), dj);
    LAUNCH_SM(k_entropy_simt, dim3((N + W - 1) / W, GEO_NSTREAM), dim3(64), (size_t)W * SB_STRIDE * 4, dj, n, (int)W);
  }
  return UVOL_OK;
}
static int geo_submit(uvol_ctx *ctx, GeoLane &L, const uvol_mesh *meshes, int n, int n_conc, bool on_device,
                      uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status, bool full);
static int geo_complete(uvol_ctx *ctx, GeoLane &L);

// First half of a group of frames on lane L: lays out the workspaces, uploads host inputs, enqueues every kernel of the group and the
// read-back of its job records.  Returns without waiting for the GPU (but for the one look at the batch's storage order, below).
// full = worst-case workspace and output sizes (the retry of a frame the compact layout could not hold); n_conc = frames of the whole
// call (what is on the chip together decides the kernel forms, not this group's share).
static int geo_submit_impl(uvol_ctx *ctx, GeoLane &L, const uvol_mesh *meshes, int n, int n_conc, bool on_device, const size_t *caps, bool full) {
  GeoState *G = ctx->geo;
  const auto t_enter = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  const uvol_params &prm = ctx->prm;
  if (prm.q_position_attr < 1 || prm.q_position_attr > 16 || prm.q_texture_attr < 1 || prm.q_texture_attr > 16 ||
      prm.q_normal_attr < 2 || prm.q_normal_attr > 16) { ctx->set_error("quantization bits out of supported range (1..16)"); return UVOL_E_UNSUPPORTED; }
  // DRACO_COMPRESSION_LEVEL only selects the encoder's tools, it is not written to the file (scripts/Encoder.py:171-179 documents
  // 0..10): every legal level is encoded with the cl-7 tool set (edgebreaker + valence contexts, parallelogram / tex-coord /
  // geometric-normal prediction, RAW rANS), which any Draco decoder reads; the shims say so on stderr
  if (prm.draco_compression_level < 0 || prm.draco_compression_level > 10) { ctx->set_error("DRACO_COMPRESSION_LEVEL %d outside 0..10", prm.draco_compression_level); return UVOL_E_INVALID; }
  // DRACO_COMPRESSION_LEVEL 0 selects sequential connectivity in stock draco_encoder (speed 10); every other level is written with the
  // level-7 tool set (valence edgebreaker)
  const bool seq = prm.draco_compression_level == 0;
  // The valence replay is one wave per frame and takes what one frame takes (~25-50 ms); the renumber / seams group it runs beside shrinks
  // with the batch.  Below ~1200 frames the replay is the longer of the two, so it is joined late (before the entropy stage) and
  // overlaps the traversals too; its inputs then cannot share bytes with the record tables (+7.7 MB per frame, irrelevant at that size).
  static const int late_env = [] { const char *e = getenv("UVOL_LATE_JOIN"); return e ? atoi(e) : -1; }();      // tests: 0 / 1 force the early / late join
  const bool late_join = late_env >= 0 ? late_env != 0 : std::max(n, n_conc) <= 1200;      // (n_conc: the frames of the whole call are on the chip together, whatever this group's share)
  L.hjobs.assign((size_t)n, GeoJob{});
  std::vector<size_t> ws_off(n), in_off(n), zero_sz(n);
  size_t ws_total = 0, in_total = 0, out_total = 0;
  uint32_t max_nfi = 0, max_vals = 0, max_ecap = 0, he_nb_max = 0, ms_nb_max = 1; bool he_part_all = true; uint64_t algo_in = 0;
  uint64_t max_ids = 0;
  for (int i = 0; i < n; i++) {
    const uvol_mesh &m = meshes[i]; max_nfi = std::max(max_nfi, m.n_faces);
    max_ids = std::max(max_ids, geo_ecap(m.n_pos, (m.uv && m.idx_uv) ? m.n_uv : 0, (m.nrm && m.idx_nrm) ? m.n_nrm : 0, m.n_faces, full));
    max_vals = std::max(max_vals, std::max(m.n_pos, std::max((m.uv && m.idx_uv) ? m.n_uv : 0u, (m.nrm && m.idx_nrm) ? m.n_nrm : 0u)));
  }
  const int r8 = geo_rec8(max_nfi, max_ids) ? 1 : 0;
  // How the serial walkers of this group run is decided here, before the workspaces are laid out: the lane-per-walker kernels read ONE
  // 16-byte record per face (format 2) where the 21-bit fields allow it, and the record tables are sized for the format
  const unsigned NC0 = (unsigned)std::max(n, n_conc);
  WalkPlan wp_walk = walk_plan(G, max_nfi, max_vals, (size_t)NC0);
  static const bool tvg_env = [] { const char *e = getenv("UVOL_TRAVERSE_VGLOBAL"); return e && *e == '1'; }();
  const bool tvg = tvg_env || ctx->prm.traverse_vbits_l2 != 0;
  WalkPlan wp_trav = walk_plan(G, max_nfi, max_vals, (size_t)3 * NC0, tvg);
  const bool f16_off = geo_rec_face_off();
  const int fmt0 = (r8 && wp_walk.simt_w && !f16_off) ? 2 : r8, fmtT = (r8 && wp_trav.simt_w && !f16_off) ? 2 : r8;
  for (int i = 0; i < n; i++) {
    const uvol_mesh &m = meshes[i]; GeoJob &J = L.hjobs[i];
    if (!m.pos || !m.idx_pos || m.n_pos == 0 || m.n_faces == 0 || m.n_faces > (1u << 26)) { ctx->set_error("mesh %d: empty or invalid", i); return UVOL_E_INVALID; }
    J.n_pos = m.n_pos; J.nf_in = m.n_faces; J.relabel = seq ? 0 : geo_relabel_mode(); J.seq = seq ? 1 : 0; J.late_join = late_join ? 1 : 0;
    J.has_uv = (m.uv && m.idx_uv && m.n_uv) ? 1 : 0; J.has_nrm = (m.nrm && m.idx_nrm && m.n_nrm) ? 1 : 0;
    J.n_uv = J.has_uv ? m.n_uv : 0; J.n_nrm = J.has_nrm ? m.n_nrm : 0;
    J.nad = J.has_uv + J.has_nrm; J.qp = prm.q_position_attr; J.qt = prm.q_texture_attr; J.qn = prm.q_normal_attr;
    { int k = 0; if (J.has_uv) J.att_kind[k++] = 0; if (J.has_nrm) J.att_kind[k++] = 1; for (; k < 2; k++) J.att_kind[k] = -1; }
    const WsPlan &wp = layout_job(J, nullptr, full, fmt0, fmtT, G->plan, G->items);
    ws_off[i] = ws_total; ws_total += wp.total; zero_sz[i] = wp.zero;
    const size_t in_sz = ((size_t)m.n_pos * 12 + 255) / 256 * 256 + ((size_t)J.n_uv * 8 + 255) / 256 * 256 + ((size_t)J.n_nrm * 12 + 255) / 256 * 256 +
                         (size_t)(1 + J.has_uv + J.has_nrm) * (((size_t)m.n_faces * 12 + 255) / 256 * 256);
    in_off[i] = in_total; in_total += on_device ? 0 : in_sz;
    // the frames' streams are packed back to back (k_out_offsets), so the batch's output area is sized for typical streams
    // (8 bytes per face; the defaults give 1.3), not for the sum of the callers' capacities; GEO_E_SLAB_FULL -> retried alone
    const size_t oc = full ? caps[i] : std::min<size_t>(caps[i], 32768 + 8 * (size_t)m.n_faces);
    out_total += (oc + 255) & ~(size_t)255; J.out_cap = (uint32_t)std::min<size_t>(caps[i], 0xffffffffu);
    max_vals = std::max(max_vals, std::max(m.n_pos, std::max(J.n_uv, J.n_nrm))); max_ecap = std::max(max_ecap, J.ecap);
    he_nb_max = std::max(he_nb_max, J.he_nb); he_part_all = he_part_all && J.he_vpb != 0;
    if (J.relabel) ms_nb_max = std::max(ms_nb_max, std::max(J.ms_nb[0], ((J.n_pos ? J.n_pos - 1 : 0) >> J.ms_sh[1]) + 1));
    algo_in += (uint64_t)m.n_pos * 12 + (uint64_t)J.n_uv * 8 + (uint64_t)J.n_nrm * 12 + (uint64_t)(1 + J.has_uv + J.has_nrm) * m.n_faces * 12;
  }
  int rc;
  if ((rc = uvol_ensure(ctx, L.slab, ws_total))) {
    // out of device memory: the workspaces idle lanes still hold from earlier (larger) groups are given back, then once more
    (void)hipGetLastError();
    for (GeoLane *o : G->lanes) if (o != &L && o->busy) { const int r = geo_complete(ctx, *o); if (r != UVOL_OK && G->deferred_rc == UVOL_OK) G->deferred_rc = r; }      // (their results first)
    for (GeoLane *o : G->lanes) if (o != &L && !o->busy) for (uvol_devbuf *b : { &o->slab, &o->inputs, &o->outs }) if (b->p) { (void)hipStreamSynchronize(o->stream); (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    if ((rc = uvol_ensure(ctx, L.slab, ws_total))) return rc;
  }
  if ((rc = uvol_ensure(ctx, L.jobs, sizeof(GeoJob) * (size_t)n))) return rc;
  if (!L.ext_out && (rc = uvol_ensure(ctx, L.outs, out_total))) return rc;
  if (!on_device && (rc = uvol_ensure(ctx, L.inputs, in_total))) return rc;
  std::vector<UvolUpItem> ups; if (!on_device) ups.reserve((size_t)n * 6);
  for (int i = 0; i < n; i++) {
    const uvol_mesh &m = meshes[i]; GeoJob &J = L.hjobs[i];
    uint8_t *base = (uint8_t *)L.slab.p + ws_off[i];
    (void)layout_job(J, base, full, fmt0, fmtT, G->plan, G->items);
    J.ws_base = base; J.ws_zero = zero_sz[i];          // cleared by ONE k_job_clear launch for the whole batch (was 2 memsets per frame)
    if (L.ext_out) { J.out_pack = L.ext_out; J.slab_cap = L.ext_cap; } else { J.out_pack = (uint8_t *)L.outs.p; J.slab_cap = out_total; }
    if (on_device) { J.pos = m.pos; J.uv = J.has_uv ? m.uv : nullptr; J.nrm = J.has_nrm ? m.nrm : nullptr; J.ipos = m.idx_pos; J.iuv = J.has_uv ? m.idx_uv : nullptr; J.inrm = J.has_nrm ? m.idx_nrm : nullptr; }
    else {
      uint8_t *ib = (uint8_t *)L.inputs.p + in_off[i]; size_t o = 0;
      auto up = [&](const void *src, size_t bytes) -> const void * {                 // queued: ONE staged upload for the whole batch below
        void *d = ib + o; if (bytes) ups.push_back(UvolUpItem{ in_off[i] + o, src, bytes }); o += (bytes + 255) / 256 * 256; return d; };
      J.pos = (const float *)up(m.pos, (size_t)m.n_pos * 12);
      J.uv = J.has_uv ? (const float *)up(m.uv, (size_t)m.n_uv * 8) : nullptr;
      J.nrm = J.has_nrm ? (const float *)up(m.nrm, (size_t)m.n_nrm * 12) : nullptr;
      J.ipos = (const uint32_t *)up(m.idx_pos, (size_t)m.n_faces * 12);
      J.iuv = J.has_uv ? (const uint32_t *)up(m.idx_uv, (size_t)m.n_faces * 12) : nullptr;
      J.inrm = J.has_nrm ? (const uint32_t *)up(m.idx_nrm, (size_t)m.n_faces * 12) : nullptr;
    }
    for (int k = 0; k < 3; k++) { J.pos_min_u[k] = 0xffffffffu; J.pos_max_u[k] = 0; }
    for (int k = 0; k < 2; k++) { J.uv_min_u[k] = 0xffffffffu; J.uv_max_u[k] = 0; J.wrap_lo[k] = 0x7fffffff; J.wrap_hi[k] = -0x7fffffff - 1; }
    // stream wiring
    for (int s = 0; s < 6; s++) J.rs[s].syms = J.ctx_sym[s];
    J.rs[6].syms = J.sym_pos; J.rs[7].syms = J.sym_uv; J.rs[8].syms = J.sym_nrm;
    J.rb[0].bits = J.start_bits; J.rb[1].bits = J.seam_bits[0]; J.rb[2].bits = J.seam_bits[1]; J.rb[3].bits = J.ori_bits; J.rb[4].bits = J.flips;
    // rabs slot 1/2 follow the attribute-data slot; slot 3 = uv orientations, slot 4 = normal flips
  }
  if (!on_device) { const int rcu = uvol_upload_staged(ctx, (uint8_t *)L.inputs.p, ups); if (rcu != UVOL_OK) return rcu; }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(L.jobs.p, L.hjobs.data(), sizeof(GeoJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  GeoJob *dj = (GeoJob *)L.jobs.p;
  const unsigned N = (unsigned)n, NC = (unsigned)std::max(n, n_conc);      // NC: frames on the chip together (all groups of the call)
  L.t_prep = ms_since(t_enter);
  if (L.producer) {                                        // the inputs are produced on the caller's stream: ordered after what it holds now, no host wait
    if (!L.ev_prod) UVOL_HIP_CHECK(ctx, hipEventCreateWithFlags(&L.ev_prod, hipEventDisableTiming));
    UVOL_HIP_CHECK(ctx, hipEventRecord(L.ev_prod, L.producer));
    UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, L.ev_prod, 0));
  }
  // The front ends (dedup, corner table: streaming kernels that fill the chip) of consecutive groups run one after the other, so that a
  // group's front end meets the WALKERS of the groups before it - latency-bound, a few hundred waves - instead of their front ends:
  // each group then gets through its bandwidth-bound phases at close to the chip's full rate and the groups stay staggered.
  static const bool fe_chain = [] { const char *e = getenv("UVOL_GEO_CHAIN"); return !(e && *e == '0'); }();
  if (fe_chain && G->fe_last && G->fe_last != L.ev_fe) UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, G->fe_last, 0));
  LAUNCH(k_job_clear, dim3(128, N), dim3(UVOL_BLOCK), dj);
  const unsigned bf = uvol_blocks(max_nfi), bc = uvol_blocks((size_t)3 * max_nfi), bv = uvol_blocks(max_vals), bci = (bc + GEO_ILP - 1) / GEO_ILP,
                 be = uvol_blocks(std::min<size_t>(max_ecap, (size_t)3 * max_nfi));       // attribute entries (<= ecap, else GEO_E_WS_OVERFLOW)
  const bool relabel = geo_relabel_on() && !seq;
  bool lockstep = true;                                      // walkers of this batch move in lock step (see below); decides lanes per wave of the traversers
  // bounding boxes first: the relabelling's Morton keys are taken over them (k_quantize uses them much later)
  LAUNCH(k_minmax, dim3(std::min(bv, 16u), N), dim3(UVOL_BLOCK), dj);
  if (seq) { const int rcq = geo_encode_sequential(ctx, dj, n, full, max_nfi, max_vals, max_ecap, algo_in); if (rcq != UVOL_OK) return rcq; UVOL_HIP_CHECK(ctx, hipEventRecord(L.ev_fe, ctx->stream)); G->fe_last = L.ev_fe; }
  else {
  {
    uvol_ctx::Scope sc(ctx, "geo.k2_dedup", algo_in);
    if (!full) {
      // UVOL_DD_SLOTS=<power of two <= 4096> (tests): LDS slots per bin; a small table forces GEO_E_DD_OVERFLOW and the retry
      uint32_t slots = DD_SLOTS; { const char *e = getenv("UVOL_DD_SLOTS"); const int v = e ? atoi(e) : 0; if (v >= 4 && v <= DD_SLOTS && !(v & (v - 1))) slots = (uint32_t)v; }
      const unsigned bt = (unsigned)((max_vals + DD_TILE - 1) / DD_TILE), nbm = (unsigned)std::min<uint64_t>(DD_MAXBINS, pow2_at_least(std::max<uint64_t>(1, max_vals / 1024)));
      LAUNCH(k_dd_count, dim3(bt, N, 3), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_dd_scan, dim3(1, N, 3), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_dd_scatter, dim3(bt, N, 3), dim3(UVOL_BLOCK), dj);
      if (max_vals / std::max(1u, nbm) <= 1100u && slots >= 2048u) LAUNCH((k_dd_resolve<2048, 4>), dim3(nbm, N, 3), dim3(UVOL_BLOCK), dj, 2048u);
      else LAUNCH((k_dd_resolve<DD_SLOTS, 6>), dim3(nbm, N, 3), dim3(UVOL_BLOCK), dj, slots);
    } else {
      LAUNCH(k_dd_clear, dim3(16, N, 3), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 0, 0);
      LAUNCH(k_dedup<2>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 1, 0);
      LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 2, 0);
      LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 0, 1);
      LAUNCH(k_dedup<2>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 1, 1);
      LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 2, 1);
    }
    const unsigned mt0 = (unsigned)((max_vals + MS_TILE - 1) / MS_TILE), mt1 = (unsigned)(((size_t)max_nfi + MS_TILE - 1) / MS_TILE);
    // How the frames are stored decides two things, so the batch is looked at once (k_coherence: one pass over the index arrays) and the
    // two counts come back to the host (the only mid-batch synchronisation; the encode kernels of a large batch take 0.2 - 0.9 s):
    //  * frames stored coherently skip the locality relabelling - if none needs it, its ~6 M (empty) workgroups are not even launched;
    //  * frames with the SAME connectivity as their predecessor (an animated mesh of fixed topology) are walked in lock step by the
    //    lanes of a wave - 16 attribute traversers per wave then beat one per wave (200 vs 300 ms per 2160 frames), while walkers on
    //    unrelated meshes diverge and miss at different times, and one per wave is the faster form (307 vs 392 ms).
    bool any_relabel = relabel;
    if (relabel || NC >= 256) {
      if (!L.counts) UVOL_HIP_CHECK(ctx, hipMalloc((void **)&L.counts, 64));
      uint32_t hc[2] = { 0, 0 };
      UVOL_HIP_CHECK(ctx, hipMemsetAsync(L.counts, 0, 64, ctx->stream));
      LAUNCH(k_coherence, dim3(bf, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_relabel_decide, dim3((N + 63) / 64), dim3(64), dj, n, L.counts);
      UVOL_HIP_CHECK(ctx, hipMemcpyAsync(hc, L.counts, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
      UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      any_relabel = relabel && hc[0] != 0;
      lockstep = (uint64_t)hc[1] * 10u >= (uint64_t)n * 9u;
    }
    if (any_relabel) {                                       // new position ids (Morton order) and the positions in that order
      LAUNCH(k_ms_key_pos, dim3(bv, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_ms_count, dim3(mt0, N), dim3(UVOL_BLOCK), dj, 0);
      LAUNCH(k_ms_scan, dim3(1, N), dim3(UVOL_BLOCK), dj, 0);
      LAUNCH(k_ms_scatter, dim3(mt0, N), dim3(UVOL_BLOCK), dj, 0);
      LAUNCH(k_ms_place, dim3(ms_nb_max, N), dim3(UVOL_BLOCK), dj, 0);
    }
    LAUNCH(k_faces, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_scan_sums, dim3(1, N), dim3(UVOL_BLOCK), dj, (int)SCAN_KEEP);
    if (any_relabel) {                                       // faces stored in the order of their lowest new vertex id
      LAUNCH(k_face_cidx, dim3(bf, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_ms_count, dim3(mt1, N), dim3(UVOL_BLOCK), dj, 1);
      LAUNCH(k_ms_scan, dim3(1, N), dim3(UVOL_BLOCK), dj, 1);
      LAUNCH(k_ms_scatter, dim3(mt1, N), dim3(UVOL_BLOCK), dj, 1);
      LAUNCH(k_ms_place, dim3(ms_nb_max, N), dim3(UVOL_BLOCK), dj, 1);
      LAUNCH(k_relabel_faces, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    }
    LAUNCH(k_compact_faces, dim3(bf, N), dim3(UVOL_BLOCK), dj);       // the frames that are not relabelled (decided per frame on the device)
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.k3_corner_table", (uint64_t)n * 0 + (uint64_t)3 * max_nfi * 4 * 3);
    if (he_part_all) {
      const unsigned bt = (unsigned)(((size_t)3 * max_nfi + HE_TILE - 1) / HE_TILE);
      LAUNCH(k_hp_count, dim3(bt, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_hp_scan, dim3(1, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_hp_scatter, dim3(bt, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_hp_build, dim3(he_nb_max, N), dim3(UVOL_BLOCK), dj);
    } else {
      LAUNCH(k_he_count, dim3(bci, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_he_scan, dim3(1, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_he_fill, dim3(bci, N), dim3(UVOL_BLOCK), dj);
    }
    LAUNCH(k_edge_match, dim3(bci, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_vert0, dim3(bv, N), dim3(UVOL_BLOCK), dj);
  }
  UVOL_HIP_CHECK(ctx, hipEventRecord(L.ev_fe, ctx->stream)); G->fe_last = L.ev_fe;
  // UVOL_SIMT_W_WALK / UVOL_SIMT_W_TRAV (diagnostic): lanes per wave of one of the two lane-per-walker kernels only (UVOL_SIMT_W sets both)
  static const int w_walk_env = [] { const char *e = getenv("UVOL_SIMT_W_WALK"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();
  static const int w_trav_env = [] { const char *e = getenv("UVOL_SIMT_W_TRAV"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();
  if (w_walk_env && wp_walk.simt_w) wp_walk.simt_w = w_walk_env;
  {
    LAUNCH(k_pack0, dim3(bf, N), dim3(UVOL_BLOCK), dj, fmt0);
    uvol_ctx::Scope sc(ctx, "geo.k4_eb_walk", (uint64_t)n * 32 * max_nfi);
    if (wp_walk.simt_w) {
      const unsigned W = (unsigned)wp_walk.simt_w, nb = (N + W - 1) / W;
      if (fmt0 == 2) { if (geo_face_bits()) LAUNCH((k_eb_walk_simt_f16<true, 0>), dim3(nb), dim3(64), dj, n, (int)W); else if (geo_walk_ld()) LAUNCH((k_eb_walk_simt_f16<false, 1>), dim3(nb), dim3(64), dj, n, (int)W); else LAUNCH((k_eb_walk_simt_f16<false, 0>), dim3(nb), dim3(64), dj, n, (int)W); }
      else if (r8) LAUNCH((k_eb_walk_simt<true>), dim3(nb), dim3(64), dj, n, (int)W); else LAUNCH((k_eb_walk_simt<false>), dim3(nb), dim3(64), dj, n, (int)W);
    }
    else if (r8) LAUNCH_SM((k_eb_walk<true>), dim3(N), dim3(128), wp_walk.lds, dj, wp_walk.vcw, geo_walk_pf());
    else LAUNCH_SM((k_eb_walk<false>), dim3(N), dim3(128), wp_walk.lds, dj, wp_walk.vcw, geo_walk_pf());
    LAUNCH(k_face_time, dim3(bf, N), dim3(UVOL_BLOCK), dj);
  }
  // valence replay + context scatter depend only on the walk: run them on the auxiliary stream, beside
  // renumber / seams / fans / DFS traversal on the main stream; joined again before the entropy stage.
  // The parallel kernels of the replay's preparation run on the MAIN stream, before the fork: beside the renumber / seams group they
  // took 25 + 39 ms of the auxiliary stream's time (its workgroups wait behind the main stream's 5 M-workgroup grids), which made
  // the auxiliary chain (125 ms) longer than the group it hides behind (70 ms) - the join below waited ~55 ms per batch.
  {
    LAUNCH(k_eb_event_flags, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_scan_sums, dim3(1, N), dim3(UVOL_BLOCK), dj, (int)SCAN_EVENTS);
    LAUNCH(k_eb_event_compact, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_valence_init, dim3(bc, N), dim3(UVOL_BLOCK), dj);
  }
  UVOL_HIP_CHECK(ctx, hipEventRecord(L.ev_walk, ctx->stream));
  UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(L.aux, L.ev_walk, 0));
  {
    { uvol_ctx::Scope sc(ctx, "geo.k4_eb_valence", 0, L.aux); LAUNCH_ON(L.aux, k_eb_valence, dim3(N), dim3(64), dj); }
    LAUNCH_ON(L.aux, k_eb_ctx, dim3(N), dim3(64), dj);
  }
  UVOL_HIP_CHECK(ctx, hipEventRecord(L.ev_val, L.aux));
  {
    uvol_ctx::Scope sc(ctx, "geo.k4b_renumber_seams", 0);
    LAUNCH(k_renumber_a, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_renumber_seams, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_scan_sums, dim3(1, N), dim3(UVOL_BLOCK), dj, (int)SCAN_ELIG);
    LAUNCH(k_seam_bits, dim3((bc + SB_E - 1) / SB_E, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_aseg_a, dim3(bci, N, 2), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_aseg_b, dim3(bci, N, 2), dim3(UVOL_BLOCK), dj);
  }
  // The auxiliary stream (events / valence replay / context scatter, ~90 ms per 2160 frames) is joined HERE, not before the
  // entropy stage: it then overlaps the renumber / seams group only (about as long), but everything it reads (old-order
  // opposite corners and vertices, the symbol sequence, the valence scratch: 11.6 MB per frame) is dead before the three record
  // tables of the attribute traversals are written and shares their addresses - the workspace peak drops from 63 to 52 MB.
  if (!late_join) UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, L.ev_val, 0));
  {
    LAUNCH(k_pack3, dim3(bf, N, 3), dim3(UVOL_BLOCK), dj, fmtT);
    uvol_ctx::Scope sc(ctx, "geo.k5_traverse", (uint64_t)n * 32 * max_nfi * 3);
    // params.traverse_vbits_l2 (or UVOL_TRAVERSE_VGLOBAL=1): LDS traversers keep only the face bitmap in LDS (25 KB -> 6 per
    // CU instead of 3), the vertex bitmap lives in L2; each walker is ~30 % slower, twice as many are resident
    if (wp_trav.simt_w > 1 && !lockstep && geo_simt_env() == 0) wp_trav.simt_w = 1;      // unrelated meshes: one traverser per wave
    if (w_trav_env && wp_trav.simt_w) wp_trav.simt_w = w_trav_env;
    launch_traversals(ctx, dj, n, wp_trav, fmtT);
  }
  { uvol_ctx::Scope sc(ctx, "geo.k5b_v2d", 0); LAUNCH(k_v2d, dim3(be, N, 3), dim3(UVOL_BLOCK), dj, fmtT); }      // (own scope: geo.k5_traverse is exactly the traversal kernel, as rocprof lists it)
  {
    uvol_ctx::Scope sc(ctx, "geo.k1_quantize", algo_in);
    LAUNCH(k_quantize, dim3(be, N, 3), dim3(UVOL_BLOCK), dj);
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.k6_predict", 0);
    LAUNCH(k_stream_setup, dim3(N), dim3(64), dj);
    LAUNCH(k_pred_pos, dim3(be, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_pred_uv, dim3(be, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_scan_blocks, dim3(be, N), dim3(UVOL_BLOCK), dj, (int)SCAN_ORI);
    LAUNCH(k_scan_sums, dim3(1, N), dim3(UVOL_BLOCK), dj, (int)SCAN_ORI);
    LAUNCH(k_ori_compact, dim3(be, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_ori_bits, dim3(be, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_face_normals, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_pred_nrm, dim3(be, N), dim3(UVOL_BLOCK), dj);
  }
  if (late_join) UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, L.ev_val, 0));
  {
    uvol_ctx::Scope sc0(ctx, "geo.k7_hist_tables", 0);
    LAUNCH(k_hist, dim3(uvol_blocks((size_t)9 * max_nfi, 16 * UVOL_BLOCK), GEO_NSTREAM, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_rans_tables, dim3(GEO_NSTREAM, N), dim3(64), dj);
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.k7_entropy_encode", 0);
    // Wave-per-stream coder (state recurrence on the scalar unit, ~90 ns per symbol) while all 14 x N one-wave workgroups are resident
    // at once (8 KiB of LDS each: 20 per CU); beyond that the lane-per-stream coder, which is slower per stream (~175 ns per symbol)
    // but runs every stream of the batch concurrently (300 frames: 28 vs 72 ms; 2160 frames: 53 ms with lanes).
    // UVOL_ENTROPY_WAVE=1 / 0 (tests / diagnostic) forces one form.
    static const int ent_env = [] { const char *e = getenv("UVOL_ENTROPY_WAVE"); return !e ? -1 : (*e == '1' ? 1 : 0); }();
    static const int ent_w_env = [] { const char *e = getenv("UVOL_ENTROPY_W"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();   // lanes per wave of the lane form (implies it)
    const bool ent_wave = ent_env >= 0 ? ent_env == 1 : (ent_w_env == 0 && (size_t)(GEO_NSTREAM + GEO_NRABS) * NC <= (size_t)20 * G->num_cu);
    if (ent_wave) LAUNCH(k_entropy_encode, dim3(N, GEO_NSTREAM + GEO_NRABS), dim3(64), dj, uvol_debug() ? 1 : 0);      // frame index fastest: the long streams of the frames spread over the four SIMDs of a CU (geom_decode.hip: k_gdec_rans)
    else {
      // lanes per wave: five streams of a frame are long (three attribute symbol streams, two seam-bit streams; ~300 k steps) and a
      // wave runs as long as its longest lane, so the launch should put at most ONE long wave on a SIMD (1024 of them): waves that
      // share a SIMD share its issue slots (2160 frames: 172 / 105 / 60 / 64 / 55 / 52 ms with 1 / 2 / 4 / 8 / 16 / 32 lanes per wave)
      unsigned W = 4; while (W < 64 && 5u * NC > 512u * W) W *= 2;
      if (ent_w_env) W = (unsigned)ent_w_env;
      LAUNCH(k_rans_recip, dim3(uvol_blocks(((size_t)2 << std::max(prm.q_position_attr, std::max(prm.q_texture_attr, prm.q_normal_attr))) + 8), GEO_NSTREAM, N), dim3(UVOL_BLOCK), dj);
      LAUNCH_SM(k_entropy_simt, dim3((N + W - 1) / W, GEO_NSTREAM + GEO_NRABS), dim3(64), (size_t)W * SB_STRIDE * 4, dj, n, (int)W);
    }
  }
  }   // !seq
  {
    uvol_ctx::Scope sc(ctx, "geo.k8_layout_gather", 0);
    if (seq) LAUNCH(k_sq_layout, dim3(N), dim3(64), dj); else LAUNCH(k_layout, dim3(N), dim3(64), dj);
    LAUNCH(k_out_offsets, dim3(1), dim3(64), dj, n);
    LAUNCH(k_gather, dim3(64, GEO_MAXPIECES, N), dim3(UVOL_BLOCK), dj);
  }
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  // (the job records are read back by geo_complete_impl: a device-to-host copy into pageable memory does not return before the stream
  // has reached it, i.e. before every kernel of this group is done - here it serialised the groups of a call)
  L.t_enq = ms_since(t_enter); L.t_enter = t_enter;
  return UVOL_OK;
}

// Second half of a group: waits for its stream, copies the packed bitstreams out (one device-to-host copy into pinned staging, then
// plain memcpy into the caller's buffers), fills out_lens / status, re-encodes the frames the compact workspace could not hold.
static int geo_complete_impl(uvol_ctx *ctx, GeoLane &L) {
  static const bool timing = [] { const char *e = getenv("UVOL_TIMING"); return e && *e == '1'; }();       // diagnostic: host-side phases of a batch on stderr
  auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  const int n = L.n; const bool full = L.full, on_device = L.on_device;
  uint8_t *const *outs = L.outp.data(); size_t *out_lens = L.out_lens; int *status = L.status;
  const auto t_enter = L.t_enter; const double t_prep = L.t_prep, t_enq = L.t_enq;
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  UVOL_HIP_CHECK(ctx, hipMemcpy(L.hjobs.data(), L.jobs.p, sizeof(GeoJob) * (size_t)n, hipMemcpyDeviceToHost));
  const double t_gpu = ms_since(t_enter);
  int worst = UVOL_OK;
  // one device-to-host copy of the packed bitstreams into pinned staging, then plain memcpy into the caller's buffers
  size_t packed = 0;
  for (int i = 0; i < n; i++) { const GeoJob &J = L.hjobs[i]; if (J.status == 0) packed = std::max<size_t>(packed, (size_t)J.out_pack_off + J.out_len); }
  if (L.ext_out) packed = 0;                               // GPU-resident form: the bitstreams stay where k_gather put them
  if (packed > L.pinned_cap) {
    if (L.pinned) (void)hipHostFree(L.pinned);
    L.pinned = nullptr; L.pinned_cap = 0;
    const size_t want = packed + packed / 4 + (1u << 20);
    UVOL_HIP_CHECK(ctx, hipHostMalloc((void **)&L.pinned, want, hipHostMallocDefault));
    L.pinned_cap = want;
  }
  if (packed) {
    UVOL_HIP_CHECK(ctx, hipMemcpyAsync(L.pinned, L.outs.p, packed, hipMemcpyDeviceToHost, ctx->stream));
    UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  const double t_d2h = ms_since(t_enter);
  std::vector<int> retry;
  // staging -> the caller's buffers: a few host threads for large batches (2160 frames x 250 KB took ~60 ms of one core per batch)
  { const int nt = packed > ((size_t)32 << 20) ? 8 : 1;
    auto copy_range = [&](int a, int b) { for (int i = a; i < b; i++) { const GeoJob &J = L.hjobs[i]; if (J.status == 0 && !L.ext_out) memcpy(outs[i], L.pinned + J.out_pack_off, J.out_len); } };
    if (nt == 1) copy_range(0, n);
    else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(copy_range, (int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt)); for (auto &x : th) x.join(); } }
  for (int i = 0; i < n; i++) {
    const GeoJob &J = L.hjobs[i];
    int st = J.status == 0 ? UVOL_OK : (J.status == UVOL_E_NOSPACE ? UVOL_E_NOSPACE : UVOL_E_ENCODE);
    out_lens[i] = J.out_len;
    if (L.ext_offs) L.ext_offs[i] = (size_t)J.out_pack_off;
    if (st == UVOL_OK) { }
    else if (!full && (J.status == GEO_E_WS_OVERFLOW || J.status == GEO_E_SLAB_FULL || J.status == GEO_E_DD_OVERFLOW)) { retry.push_back(i); st = UVOL_OK; }
    else { ctx->set_error("mesh %d: encode failed (device status %d)", i, J.status); worst = st; }
    if (status) status[i] = st;
  }
  // frames the compact workspace (or the packed output area) could not hold: once more, alone, with worst-case sizes
  if (!retry.empty()) {
    // the lane's own record of the group is replaced by the one-frame retries: keep what is still needed
    const std::vector<uvol_mesh> rm(L.meshes); const std::vector<uint8_t *> ro(L.outp); const std::vector<size_t> rcap(L.caps);
    // GPU-resident form: a retried frame goes behind the frames already packed into the caller's buffer
    uint8_t *const ext0 = L.ext_out; const size_t ext_cap0 = L.ext_cap; size_t *const offs0 = L.ext_offs; size_t tail = 0;
    if (ext0) for (int i = 0; i < n; i++) { const GeoJob &J = L.hjobs[i]; if (J.status == 0) tail = std::max<size_t>(tail, ((size_t)J.out_pack_off + J.out_len + 255) & ~(size_t)255); }
    for (int i : retry) {
      if (timing) fprintf(stderr, "[uvol-timing] mesh %d: re-encoding with worst-case workspace\n", i);
      int st1 = UVOL_OK;
      if (ext0) {
        if (tail >= ext_cap0) { if (status) status[i] = UVOL_E_NOSPACE; worst = UVOL_E_NOSPACE; out_lens[i] = 0; continue; }
        L.ext_out = ext0 + tail; L.ext_cap = ext_cap0 - tail; L.ext_offs = nullptr; L.producer = nullptr;
      }
      size_t cap1 = ext0 ? std::min<size_t>(ext_cap0 - tail, 0xffffffffu) : rcap[i];
      int rc1 = geo_submit(ctx, L, &rm[i], 1, 1, on_device, &ro[i], &cap1, out_lens + i, &st1, true);
      if (rc1 == UVOL_OK) rc1 = geo_complete(ctx, L);
      if (ext0) { L.ext_out = ext0; L.ext_cap = ext_cap0; L.ext_offs = offs0; if (offs0) offs0[i] = tail; if (rc1 == UVOL_OK && st1 == UVOL_OK) tail = (tail + out_lens[i] + 255) & ~(size_t)255; }
      if (rc1 != UVOL_OK) return rc1;
      if (status) status[i] = st1;
      if (st1 != UVOL_OK) worst = st1;
    }
  }
  if (timing) fprintf(stderr, "[uvol-timing] geo group n=%d sizeof(GeoJob)=%zu: host prepared %.1f ms, enqueued %.1f, gpu done %.1f, packed d2h %.1f, copied out %.1f (enter at %.1f)\n", n, sizeof(GeoJob), t_prep, t_enq, t_gpu, t_d2h, ms_since(t_enter),
                      std::chrono::duration<double, std::milli>(t_enter.time_since_epoch()).count());
  return status ? UVOL_OK : worst;
}

// the lane's streams stand in for the context's while one of its groups is submitted / completed (LAUNCH, Scope, uvol_ensure use ctx->stream)
static int geo_submit(uvol_ctx *ctx, GeoLane &L, const uvol_mesh *meshes, int n, int n_conc, bool on_device,
                      uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status, bool full) {
  L.meshes.assign(meshes, meshes + n); L.outp.assign(outs, outs + n); L.caps.assign(caps, caps + n);
  L.out_lens = out_lens; L.status = status; L.n = n; L.n_conc = n_conc; L.on_device = on_device; L.full = full;
  hipStream_t saved = ctx->stream; ctx->stream = L.stream;
  const int rc = geo_submit_impl(ctx, L, L.meshes.data(), n, n_conc, on_device, L.caps.data(), full);
  ctx->stream = saved;
  L.busy = rc == UVOL_OK;
  return rc;
}
static int geo_complete(uvol_ctx *ctx, GeoLane &L) {
  if (!L.busy) return UVOL_OK;
  L.busy = false;
  hipStream_t saved = ctx->stream; ctx->stream = L.stream;
  const int rc = geo_complete_impl(ctx, L);
  ctx->stream = saved;
  return rc;
}

// completes every group still in flight (enqueued calls complete lazily, so that the next call's front end overlaps this call's walkers);
// returns the first error among them and among the groups completed earlier on behalf of later calls
int geo_flush(uvol_ctx *ctx) {
  GeoState *G = ctx->geo; if (!G) return UVOL_OK;
  int rc = G->deferred_rc; G->deferred_rc = UVOL_OK;
  const int nl = (int)G->lanes.size();
  for (int k = 0; k < nl; k++) { GeoLane *L = G->lanes[(G->next_lane + k) % nl]; const int r = geo_complete(ctx, *L); if (rc == UVOL_OK) rc = r; }
  ctx->resolve_profile();
  return rc;
}

// lanes of a context: UVOL_GEO_LANES (default 2; 1 = every call one group on the context's stream, as before round 4).  Measured on 2560
// distinct frames per call, enqueued calls: 1 / 2 / 3 / 4 lanes = 3176 / 3433 / 3401 / 3420 frames/s geometry alone, 2607 / 2682 / 2421 / 2517
// beside the texture context - two groups overlap their front ends and walkers, more only add interference - while a blocking call cut
// into four groups was SLOWER than one group (2454 against 3176: nothing runs beside the last group's walkers, and every group pays
// its own read-back).
static inline int geo_lanes_wanted() { static const int v = [] { const char *e = getenv("UVOL_GEO_LANES"); const int k = e ? atoi(e) : 2; return k < 1 ? 1 : (k > 16 ? 16 : k); }(); return v; }
// frames per group at least (UVOL_GEO_MIN_GROUP, tests: small values spread small calls over the lanes): below 2 x this a call stays one
// group - its walkers are the whole critical path anyway
static inline int geo_min_group() { static const int v = [] { const char *e = getenv("UVOL_GEO_MIN_GROUP"); const int k = e ? atoi(e) : 160; return k < 1 ? 1 : k; }(); return v; }

// Enqueue n frames: the call is cut into up to `lanes` contiguous groups, each submitted on the next lane of the ring.  A lane that still
// holds a group of an EARLIER call is completed first (its error, if any, is kept for geo_flush).
// split: cut the call into groups (enqueued calls, whose successor overlaps their tail; blocking calls with HOST inputs, whose groups
// upload while the groups before them encode); a blocking call on device inputs stays one group.
int geo_encode_batch_begin(uvol_ctx *ctx, const uvol_mesh *meshes, int n, bool on_device,
                           uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status, bool split) {
  GeoState *G = ctx->geo;
  if (n <= 0) return UVOL_OK;
  // host inputs: four groups at least (a group uploads while the groups before it encode; the first group's upload is the only one nothing hides)
  const int want = on_device ? geo_lanes_wanted() : std::max(geo_lanes_wanted(), 4);
  static const int split_env = [] { const char *e = getenv("UVOL_GEO_SPLIT"); return e ? atoi(e) : -1; }();      // tests / diagnostic: 1 / 0 force / forbid the split
  if (split_env >= 0) split = split_env != 0;
  const int groups = split ? std::max(1, std::min(want, n / geo_min_group())) : 1;
  for (int g = 0; g < groups; g++) {
    const int a = (int)((long long)n * g / groups), b = (int)((long long)n * (g + 1) / groups);
    // a blocking call on device inputs always runs on lane 0 (one workspace of its size per context, as before); the others take the ring
    GeoLane *L = geo_lane(ctx, split ? G->next_lane % want : 0);
    if (!L) { ctx->set_error("geometry lane: stream / event creation failed"); return UVOL_E_HIP; }
    if (split) G->next_lane = (G->next_lane + 1) % want;
    if (L->busy) { const int r = geo_complete(ctx, *L); if (r != UVOL_OK && G->deferred_rc == UVOL_OK) G->deferred_rc = r; }
    const int rc = geo_submit(ctx, *L, meshes + a, b - a, n, on_device, outs + a, caps + a, out_lens + a, status ? status + a : nullptr, false);
    if (rc != UVOL_OK) return rc;
  }
  return UVOL_OK;
}
// GPU-resident form: one group on lane 0 (the packed output area is one caller buffer), ordered after `producer`
int geo_encode_batch_dev_out(uvol_ctx *ctx, const uvol_mesh *meshes, int n, hipStream_t producer, uint8_t *dev_out, size_t dev_cap, size_t *out_offs, size_t *out_lens, int *status) {
  if (n <= 0) return UVOL_OK;
  int rf = geo_flush(ctx);                                 // nothing else of this context in flight
  GeoLane *L = geo_lane(ctx, 0);
  if (!L) { ctx->set_error("geometry lane: stream / event creation failed"); return UVOL_E_HIP; }
  std::vector<uint8_t *> outs((size_t)n, nullptr); std::vector<size_t> caps((size_t)n, std::min<size_t>(dev_cap, 0xffffffffu));
  L->ext_out = dev_out; L->ext_cap = dev_cap; L->ext_offs = out_offs; L->producer = producer;
  int rc = geo_submit(ctx, *L, meshes, n, n, true, outs.data(), caps.data(), out_lens, status, false);
  if (rc == UVOL_OK) rc = geo_complete(ctx, *L);
  L->ext_out = nullptr; L->ext_cap = 0; L->ext_offs = nullptr; L->producer = nullptr;
  ctx->resolve_profile();
  return rf != UVOL_OK ? rf : rc;
}
// blocking form: begin + flush
int geo_encode_batch(uvol_ctx *ctx, const uvol_mesh *meshes, int n, bool on_device,
                     uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status) {
  const int rc = geo_encode_batch_begin(ctx, meshes, n, on_device, outs, caps, out_lens, status, !on_device);
  const int rf = geo_flush(ctx);
  return rc != UVOL_OK ? rc : rf;
}
