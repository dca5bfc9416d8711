// geo_seq.hpp - sequential connectivity (DRACO_COMPRESSION_LEVEL 0).
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
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

