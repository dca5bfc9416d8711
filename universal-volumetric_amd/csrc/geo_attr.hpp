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
// Quantisation BY VALUE ID (round 5): every position / texture coordinate / normal of the input arrays is quantised once, where it lies,
// into 16-bit fields (quantisation bits <= 16: uvol_params) - one 8-byte record per position, 4 bytes per texture coordinate and per
// octahedral normal.  The predictors gather these by the value ids of the stored corner table; until round 4 a k_quantize pass wrote the
// entries' values in coding order (P / U / O: 17 MB of HBM traffic per frame, after the traversals, on the critical chain) and every
// predictor operand went corner -> vertex -> coding order -> value.  A value no entry refers to is quantised for nothing; the wrap
// transform's bounds, which run over the CODED values only, are taken by k_v2d.  grid.z selects the attribute: 0 position, 1 uv, 2 normal.
__global__ void __launch_bounds__(UVOL_BLOCK) k_quant_ids(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t id = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const int a = blockIdx.z;
  if (a == 0) {
    if (id >= J.n_pos) return;
    const float range = quant_range(J.pos_min_u, J.pos_max_u, 3), inv = (float)((1u << J.qp) - 1) / range;
    const float *v = (J.relabel ? J.pos_s : J.pos) + 3 * (size_t)id;
    uint32_t q[3];
    for (int k = 0; k < 3; k++) { float t = v[k] - g_float_unorder(J.pos_min_u[k]); t = t * inv; q[k] = (uint32_t)(int)floorf(t + 0.5f) & 0xffffu; }
    reinterpret_cast<uint2 *>(J.qpos)[id] = make_uint2(q[0] | (q[1] << 16), q[2]);
  } else if (a == 1) {
    if (!J.has_uv || id >= J.n_uv) return;
    const float range = quant_range(J.uv_min_u, J.uv_max_u, 2), inv = (float)((1u << J.qt) - 1) / range;
    const float *v = J.uv + 2 * (size_t)id;
    uint32_t q[2];
    for (int k = 0; k < 2; k++) { float t = v[k] - g_float_unorder(J.uv_min_u[k]); t = t * inv; q[k] = (uint32_t)(int)floorf(t + 0.5f) & 0xffffu; }
    reinterpret_cast<uint32_t *>(J.quv)[id] = q[0] | (q[1] << 16);
  } else {
    if (!J.has_nrm || id >= J.n_nrm) return;
    GOct ot = g_oct(J.qn); int s, tt;
    float_to_oct(ot, J.nrm + 3 * (size_t)id, s, tt);
    reinterpret_cast<uint32_t *>(J.qnrm)[id] = ((uint32_t)s & 0xffffu) | ((uint32_t)tt << 16);
  }
}
struct GQ3 { int x, y, z; };
__device__ __forceinline__ GQ3 gq_pos(const GeoJob &J, int pid) { const uint2 r = reinterpret_cast<const uint2 *>(J.qpos)[pid]; GQ3 q; q.x = (int)(r.x & 0xffffu); q.y = (int)(r.x >> 16); q.z = (int)(r.y & 0xffffu); return q; }
__device__ __forceinline__ void gq_uv(const GeoJob &J, int uid, long long o[2]) { const uint32_t r = reinterpret_cast<const uint32_t *>(J.quv)[uid]; o[0] = (long long)(r & 0xffffu); o[1] = (long long)(r >> 16); }
// the table attribute slot i is sequenced by: its own (interior seams) or the base table
__device__ inline void attr_order(const GeoJob &J, int i, const int32_t *&order, const int32_t *&v2d, uint32_t &ne) {
  if (J.interior_seams[i]) { order = J.order[1 + i]; v2d = J.v2d[1 + i]; ne = J.ne[1 + i]; }
  else { order = J.order[0]; v2d = J.v2d[0]; ne = J.ne[0]; }
}

// ------------------------------------------------------------------------------------------------
// K6: prediction residuals - parallel per entry (all originals are known on the encoder side).  order[] holds corners of the STORED
// table; the quantised value of the entry at a corner is the quantised value of the corner's value id (every corner of a vertex -
// of an attribute vertex: of a seam-free segment of its fan - carries the same id), so no operand goes through the coding order:
// v2d[] is only asked "was this vertex coded before entry p?".
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(UVOL_BLOCK) k_pred_pos(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t p = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (p >= J.ne[0]) return;
  const int32_t *v2d = J.v2d[0];
  const int ci = J.order[0][p];
  const GQ3 own = gq_pos(J, J.cp[ci]);
  long long pred[3] = {0, 0, 0};
  if (p > 0) {
    bool have = false;
    const int oci = J.opp[ci];
    if (oci >= 0) {
      const int fo = 3 * (oci / 3), j = oci - fo;
      const uvol_s3 v3 = *reinterpret_cast<const uvol_s3 *>(geo_vt(J) + fo);
      const int vv[3] = { v3.x, v3.y, v3.z };
      const uint32_t a = (uint32_t)v2d[vv[j]], bn = (uint32_t)v2d[vv[(j + 1) % 3]], bp = (uint32_t)v2d[vv[(j + 2) % 3]];
      if (a < p && bn < p && bp < p) {
        const uvol_s3 c3 = J.extra_v ? *reinterpret_cast<const uvol_s3 *>(J.cp + fo) : v3;       // (no non-manifold vertex: vertex ids ARE position ids)
        const int cc[3] = { c3.x, c3.y, c3.z };
        const GQ3 qa = gq_pos(J, cc[j]), qn = gq_pos(J, cc[(j + 1) % 3]), qp = gq_pos(J, cc[(j + 2) % 3]);
        pred[0] = (long long)qn.x + qp.x - qa.x; pred[1] = (long long)qn.y + qp.y - qa.y; pred[2] = (long long)qn.z + qp.z - qa.z;
        have = true;
      }
    }
    if (!have) { const GQ3 q1 = gq_pos(J, J.cp[J.order[0][p - 1]]); pred[0] = q1.x; pred[1] = q1.y; pred[2] = q1.z; }
  }
  const int o[3] = { own.x, own.y, own.z };
  for (int k = 0; k < 3; k++) J.sym_pos[3 * p + k] = g_sym_of(g_wrap_corr(J.wrap_lo[0], J.wrap_hi[0], o[k], pred[k]));
}

__global__ void __launch_bounds__(UVOL_BLOCK) k_pred_uv(GeoJob *jobs) {
  JOB_OR_RETURN;
  int i = -1; for (int k = 0; k < J.nad; k++) if (J.att_kind[k] == 0) i = k;
  if (i < 0) return;
  const int32_t *order, *v2d; uint32_t ne; attr_order(J, i, order, v2d, ne);
  const uint32_t p = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (p >= ne) return;
  const int c = order[p], f3 = 3 * (c / 3), j = c - f3, cnx = f3 + (j + 1) % 3, cpv = f3 + (j + 2) % 3;
  const uvol_s3 u3 = *reinterpret_cast<const uvol_s3 *>(J.cu + f3);
  const int uu[3] = { u3.x, u3.y, u3.z };
  const uint32_t nd = (uint32_t)v2d[att_vertex(J, i, cnx)], pd = (uint32_t)v2d[att_vertex(J, i, cpv)];
  long long own[2], nuv[2], puv[2];
  gq_uv(J, uu[j], own);
  long long pred[2] = {0, 0}; bool have = false; uint8_t has_ori = 0, ori = 0;
  if (nd < p) gq_uv(J, uu[(j + 1) % 3], nuv);
  if (pd < p && nd < p) {
    gq_uv(J, uu[(j + 2) % 3], puv);
    if (puv[0] == nuv[0] && puv[1] == nuv[1]) { pred[0] = puv[0]; pred[1] = puv[1]; have = true; }
    else {
      const uvol_s3 c3 = *reinterpret_cast<const uvol_s3 *>(J.cp + f3);
      const int cc[3] = { c3.x, c3.y, c3.z };
      const GQ3 qt = gq_pos(J, cc[j]), qn = gq_pos(J, cc[(j + 1) % 3]), qp = gq_pos(J, cc[(j + 2) % 3]);
      const long long tip[3] = { qt.x, qt.y, qt.z }, np_[3] = { qn.x, qn.y, qn.z }, pp_[3] = { qp.x, qp.y, qp.z };
      long long pn[3], pn2 = 0, dd = 0;
      for (int k = 0; k < 3; k++) { pn[k] = pp_[k] - np_[k]; pn2 += pn[k] * pn[k]; }
      if (pn2 != 0) {
        for (int k = 0; k < 3; k++) dd += pn[k] * (tip[k] - np_[k]);
        const long long pnuv[2] = { puv[0] - nuv[0], puv[1] - nuv[1] };
        const long long xuv[2] = { nuv[0] * pn2 + dd * pnuv[0], nuv[1] * pn2 + dd * pnuv[1] };
        long long cx2 = 0;
        for (int k = 0; k < 3; k++) { long long xp = np_[k] + (dd * pn[k]) / pn2; long long e = tip[k] - xp; cx2 += e * e; }
        const long long ns_ = (long long)g_isqrt((uint64_t)cx2 * (uint64_t)pn2);
        const long long cxuv[2] = { pnuv[1] * ns_, -pnuv[0] * ns_ };
        const long long p0[2] = { (xuv[0] + cxuv[0]) / pn2, (xuv[1] + cxuv[1]) / pn2 }, p1[2] = { (xuv[0] - cxuv[0]) / pn2, (xuv[1] - cxuv[1]) / pn2 };
        const long long cu0 = own[0], cu1 = own[1];
        const long long d0 = (cu0 - p0[0]) * (cu0 - p0[0]) + (cu1 - p0[1]) * (cu1 - p0[1]);
        const long long d1 = (cu0 - p1[0]) * (cu0 - p1[0]) + (cu1 - p1[1]) * (cu1 - p1[1]);
        has_ori = 1;
        if (d0 < d1) { pred[0] = p0[0]; pred[1] = p0[1]; ori = 1; } else { pred[0] = p1[0]; pred[1] = p1[1]; ori = 0; }
        have = true;
      }
    }
  }
  if (!have) {
    if (nd < p) { pred[0] = nuv[0]; pred[1] = nuv[1]; }
    else if (p > 0) { long long q1[2]; gq_uv(J, J.cu[order[p - 1]], q1); pred[0] = q1[0]; pred[1] = q1[1]; }
  }
  J.has_ori[p] = has_ori; J.ori_val[p] = ori;
  for (int k = 0; k < 2; k++) J.sym_uv[2 * p + k] = g_sym_of(g_wrap_corr(J.wrap_lo[1], J.wrap_hi[1], (int)own[k], (long long)(int)pred[k]));
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
// the face's un-normalised normal, the same whichever of its corners the fan walk arrives at.  It is computed once per STORED face here
// (three 8-byte quantised positions by id) instead of once per face AND vertex inside the walk, which then gathers one 12- / 24-byte
// normal per face (round 4: k_pred_nrm 53 -> 15.3 MB of HBM traffic per frame).
__global__ void __launch_bounds__(UVOL_BLOCK) k_face_normals(GeoJob *jobs) {
  JOB_OR_RETURN;
  if (!J.has_nrm) return;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f >= J.nf) return;
  const uvol_s3 c3 = *reinterpret_cast<const uvol_s3 *>(J.cp + 3 * (size_t)f);
  const GQ3 q0 = gq_pos(J, c3.x), q1 = gq_pos(J, c3.y), q2 = gq_pos(J, c3.z);
  const long long dn[3] = { (long long)q1.x - q0.x, (long long)q1.y - q0.y, (long long)q1.z - q0.z }, dp[3] = { (long long)q2.x - q0.x, (long long)q2.y - q0.y, (long long)q2.z - q0.z };
  const long long n0 = dn[1] * dp[2] - dn[2] * dp[1], n1 = dn[2] * dp[0] - dn[0] * dp[2], n2 = dn[0] * dp[1] - dn[1] * dp[0];
  // |components| < 2^(2 qp + 1): three 32-bit words per face up to 15 bits of quantisation (12 bytes per face), 64-bit words for 16
  if (J.qp <= 15) { uvol_s3 o; o.x = (int32_t)n0; o.y = (int32_t)n1; o.z = (int32_t)n2; *reinterpret_cast<uvol_s3 *>(reinterpret_cast<int32_t *>(J.fnorm) + 3 * (size_t)f) = o; }
  else { long long *o = J.fnorm + 3 * (size_t)f; o[0] = n0; o[1] = n1; o[2] = n2; }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_pred_nrm(GeoJob *jobs) {
  JOB_OR_RETURN;
  int i = -1; for (int k = 0; k < J.nad; k++) if (J.att_kind[k] == 1) i = k;
  if (i < 0) return;
  const int32_t *order, *v2d; uint32_t ne; attr_order(J, i, order, v2d, ne);
  const uint32_t d = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (d == 0) J.rb[4].n = ne;
  if (d >= ne) return;
  const int slot = J.interior_seams[i] ? i : -1;                          // the fan is cut at the attribute's seams
  const long long *FN = J.fnorm;
  const GOct ot = g_oct(J.qn);
  const int c0 = order[d];
  long long N[3] = {0, 0, 0};
  int c = c0; bool left = true; uint32_t guard = 0;
  while (c >= 0 && guard++ <= J.nc) {
    // (the face's normal, whichever corner of it c is: k_face_normals)
    if (J.qp <= 15) { const uvol_s3 fn = *reinterpret_cast<const uvol_s3 *>(reinterpret_cast<const int32_t *>(FN) + 3 * (size_t)(c / 3)); N[0] += fn.x; N[1] += fn.y; N[2] += fn.z; }
    else { const long long *fn = FN + 3 * (size_t)(c / 3); N[0] += fn[0]; N[1] += fn[1]; N[2] += fn[2]; }
    if (left) { c = st_swl(J, slot, c); if (c == c0) break; if (c < 0) { left = false; c = st_swr(J, slot, c0); } }
    else c = st_swr(J, slot, c);
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
  const uint32_t oq = reinterpret_cast<const uint32_t *>(J.qnrm)[J.cn[c0]];
  const int orig[2] = { (int)(oq & 0xffffu), (int)(oq >> 16) };
  g_oct_corr(ot, orig, ppos, cpos); g_oct_corr(ot, orig, pneg, cneg);
  for (int k = 0; k < 2; k++) { cpos[k] = g_modmax(ot, cpos[k]); cneg[k] = g_modmax(ot, cneg[k]); }
  const int *ch; uint8_t flip;
  if (g_iabs(cpos[0]) + g_iabs(cpos[1]) < g_iabs(cneg[0]) + g_iabs(cneg[1])) { flip = 0; ch = cpos; } else { flip = 1; ch = cneg; }
  J.flips[d] = flip;
  if (!flip) atomicAdd(&J.rb[4].zeros, 1u);
  for (int k = 0; k < 2; k++) J.sym_nrm[2 * d + k] = (uint32_t)(ch[k] < 0 ? ch[k] + ot.MAXQ : ch[k]);
}
