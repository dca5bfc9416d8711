// geo_attr.hpp - K1 / K6: min/max, quantisation, prediction residuals.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
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

