// geo_faces.hpp - locality relabelling, canonical faces, degenerate-face compaction.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
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
  uint32_t share = 0, tight = 0, same = 0, degen = 0;
  if (on && f < J.nf_in) {
    const uint32_t a0 = J.ipos[3 * f], a1 = J.ipos[3 * f + 1], a2 = J.ipos[3 * f + 2];
    degen = (a0 == a1 || a1 == a2 || a0 == a2) ? 1u : 0u;       // (a dropped face, as long as no two positions are equal: k_relabel_decide)
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
  const uint32_t s1 = block_sum(share), s2 = block_sum(tight), s3 = block_sum(same), s4 = block_sum(degen);
  if (threadIdx.x == 0 && on) { if (s1) atomicAdd(&J.coh_share, s1); if (s2) atomicAdd(&J.coh_tight, s2); if (s3) atomicAdd(&J.coh_same, s3); if (s4) atomicAdd(&J.n_degen, s4); }
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
  // frames the compact layout cannot hold: relabelled ones, and ones whose stored value ids are not the caller's index arrays
  if (J.status == 0 && (J.relabel || J.n_degen || J.n_dup[0] || (J.has_uv && J.n_dup[1]) || (J.has_nrm && J.n_dup[2]))) atomicAdd(&counts[2], 1u);
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
// The stored corner table's value ids.  When no face is dropped and an attribute has no two equal values - a clean export, the usual
// case - the canonical ids of that attribute ARE the caller's index array: the job's pointer is turned to it (one thread) and the
// 2.4 MB copy per attribute and 200 k-face frame is neither written nor read back from its own address (UVOL_FACE_ALIAS=0, tests:
// always copy).  Nothing downstream writes cp / cu / cn.
__global__ void __launch_bounds__(UVOL_BLOCK) k_compact_faces(GeoJob *jobs, int alias_ok) {
  GeoJob &J = jobs[blockIdx.y];
  if (J.relabel) return;                                   // block-uniform: k_face_cidx + k_relabel_faces store the faces instead
  uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  bool live = J.status == 0 && f < J.nf_in;
  const bool all_kept = alias_ok && J.status == 0 && J.bsum[uvol_blocks_dev(J.nf_in)] == J.nf_in;      // (uniform: written by k_scan_sums)
  const bool al0 = all_kept && !J.n_dup[0], al1 = all_kept && J.has_uv && !J.n_dup[1], al2 = all_kept && J.has_nrm && !J.n_dup[2];
  uint32_t v = live ? J.keep[f] : 0, tot;
  uint32_t pos = block_excl_scan(v, &tot) + (blockIdx.x <= uvol_blocks_dev(J.nf_in) ? J.bsum[blockIdx.x] : 0);
  if (live && v && !J.compact) {                          // one 12-byte store per array and face instead of three dword stores
    uvol_s3 a, b, c;
    b.x = b.y = b.z = 0; c.x = c.y = c.z = 0;
    if (!al0) { a.x = (int32_t)J.canon[0][J.ipos[3 * f]]; a.y = (int32_t)J.canon[0][J.ipos[3 * f + 1]]; a.z = (int32_t)J.canon[0][J.ipos[3 * f + 2]]; *reinterpret_cast<uvol_s3 *>(J.cp + 3 * (size_t)pos) = a; }
    if (J.has_uv && !al1) { b.x = (int32_t)J.canon[1][J.iuv[3 * f]]; b.y = (int32_t)J.canon[1][J.iuv[3 * f + 1]]; b.z = (int32_t)J.canon[1][J.iuv[3 * f + 2]]; }
    if (J.has_nrm && !al2) { c.x = (int32_t)J.canon[2][J.inrm[3 * f]]; c.y = (int32_t)J.canon[2][J.inrm[3 * f + 1]]; c.z = (int32_t)J.canon[2][J.inrm[3 * f + 2]]; }
    if (!al1) *reinterpret_cast<uvol_s3 *>(J.cu + 3 * (size_t)pos) = b;
    if (!al2) *reinterpret_cast<uvol_s3 *>(J.cn + 3 * (size_t)pos) = c;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) {
    uint32_t nf = J.bsum[uvol_blocks_dev(J.nf_in)];
    J.nf = nf; J.nc = 3 * nf;
    if (nf == 0) J.status = -3;
    if (J.compact && !(al0 && (al1 || !J.has_uv) && (al2 || !J.has_nrm))) J.status = -23;      // (cannot happen: the host lays such a group out again)
    if (al0) J.cp = reinterpret_cast<int32_t *>(const_cast<uint32_t *>(J.ipos));
    if (al1) J.cu = reinterpret_cast<int32_t *>(const_cast<uint32_t *>(J.iuv));
    if (al2) J.cn = reinterpret_cast<int32_t *>(const_cast<uint32_t *>(J.inrm));
  }
}

