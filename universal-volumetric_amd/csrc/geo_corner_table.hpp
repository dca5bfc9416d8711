// geo_corner_table.hpp - K3: opposite corners (half-edge buckets) and vertex ids.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
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
  if ((uint32_t)cnt == n) {                                             // one fan: the position is the vertex, cp[] of its corners is their vertex id (geo_vt)
    J.vopen_d[0][p] = open ? 1 : 0; J.ring_d[p] = (int32_t)(open ? n + 1 : n);
    atomicAdd(&J.nverts, 1u);
    return;
  }
  atomicOr(&J.nmbits[p >> 5], 1u << (p & 31));                          // non-manifold vertex: one id per fan, written to vert[]
  bool first = true;
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
// Vertex id per corner.  A frame without a non-manifold position (extra_v == 0: the usual case) has vert == cp and nothing is written
// or read under that name - every consumer asks geo_vt().  Otherwise k_vert0 has written the corners of the non-manifold positions
// and this pass fills in the rest, so that vert[] is complete.
__device__ __forceinline__ const int32_t *geo_vt(const GeoJob &J) { return J.extra_v ? J.vert : J.cp; }
__global__ void __launch_bounds__(UVOL_BLOCK) k_vert_fill(GeoJob *jobs) {
  JOB_OR_RETURN;
  if (J.extra_v == 0) return;                                           // block-uniform
  const uint32_t c = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (c >= J.nc) return;
  const uint32_t p = (uint32_t)J.cp[c];
  if (!((J.nmbits[p >> 5] >> (p & 31)) & 1u)) J.vert[c] = (int32_t)p;
}
