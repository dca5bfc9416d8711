// geo_dedup.hpp - K2: bitwise value dedup (hash-table and partitioned forms).
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
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
  if (threadIdx.x == 0) J.n_dup[which] = 1;               // (k_compact_faces: the canonical ids of this attribute are not the input's own)
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
