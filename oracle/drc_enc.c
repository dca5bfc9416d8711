/* oracle/drc_enc.c — TEST INFRASTRUCTURE (see oracle_common.h).
 * CPU restatement of `draco_encoder -qp Q -qt Q -qn Q -cl 7` (scripts/Encoder.py:260):
 * value dedup -> corner table -> valence-edgebreaker connectivity -> attribute seams ->
 * depth-first attribute sequencing -> quantization -> parallelogram / tex-coord-portable /
 * geometric-normal prediction -> wrap / canonicalised-octahedron transforms -> rANS (RAW).
 * The arithmetic origin is google/draco (not vendored in /root/reference); this file restates its
 * published encoder behaviour (SURVEY.md A.10 + the decoder-side A.3-A.9 it must invert) and is
 * validated by round-tripping through drc_dec.c, which the reference fixtures pin.
 *
 * Vertex identity: a base-table vertex is a *fan* of corners; its id is the fan's canonical corner
 * (left-most corner of an open fan, minimum corner id of a closed one).  Traversal output depends
 * only on vertex identity, never on numbering, so this is equivalent to draco's dense ids.
 */
#include "drc_internal.h"
#include <stdio.h>

/* ---------------- small helpers ---------------- */
static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

/* canon[i] = smallest j with identical bytes */
static void dedup_values(const void *data, uint32_t n, size_t stride, uint32_t *canon) {
  uint32_t cap = 16; while (cap < 2 * n + 2) cap *= 2;
  uint32_t *tab = (uint32_t *)malloc(4 * (size_t)cap); memset(tab, 0xff, 4 * (size_t)cap);
  const uint8_t *p = (const uint8_t *)data;
  for (uint32_t i = 0; i < n; i++) {
    uint64_t h = 1469598103934665603ULL;
    for (size_t k = 0; k < stride; k += 4) { uint32_t w; memcpy(&w, p + i * stride + k, 4); h = mix64(h ^ w); }
    uint32_t s = (uint32_t)h & (cap - 1);
    for (;;) {
      if (tab[s] == 0xffffffffu) { tab[s] = i; canon[i] = i; break; }
      if (!memcmp(p + (size_t)tab[s] * stride, p + (size_t)i * stride, stride)) { canon[i] = tab[s]; break; }
      s = (s + 1) & (cap - 1);
    }
  }
  free(tab);
}

/* directed-edge table: key (a,b) -> min corner whose opposite edge runs a->b */
typedef struct { uint64_t *key; int32_t *val; uint32_t cap; } emap;
static void emap_init(emap *m, uint32_t n) { m->cap = 16; while (m->cap < 2 * n + 2) m->cap *= 2; m->key = (uint64_t *)malloc(8 * (size_t)m->cap); m->val = (int32_t *)malloc(4 * (size_t)m->cap); memset(m->key, 0xff, 8 * (size_t)m->cap); }
static void emap_put_min(emap *m, uint64_t k, int32_t v) {
  uint32_t s = (uint32_t)mix64(k) & (m->cap - 1);
  for (;;) { if (m->key[s] == ~0ULL) { m->key[s] = k; m->val[s] = v; return; } if (m->key[s] == k) { if (v < m->val[s]) m->val[s] = v; return; } s = (s + 1) & (m->cap - 1); }
}
static int32_t emap_get(const emap *m, uint64_t k) {
  uint32_t s = (uint32_t)mix64(k) & (m->cap - 1);
  for (;;) { if (m->key[s] == ~0ULL) return -1; if (m->key[s] == k) return m->val[s]; s = (s + 1) & (m->cap - 1); }
}

/* fans of a (possibly seam-masked) corner table: vert[c] = canonical corner, open[v], ring[v] = #ring vertices */
static int compute_fans(int nf, const int32_t *opp, const uint8_t *seam, int32_t *vert, uint8_t *open, int32_t *ring) {
  ctab T = { nf, 3 * nf, opp, seam, NULL, NULL };
  int nc = 3 * nf, count = 0;
  for (int c = 0; c < nc; c++) vert[c] = -1;
  for (int c = 0; c < nc; c++) {
    if (vert[c] != -1) continue;
    int l = c, closed = 0;
    for (;;) { int nl = t_swing_left(&T, l); if (nl < 0) break; if (nl == c) { closed = 1; break; } l = nl; }
    if (closed) {
      int mn = c, k = 0; for (int a = c;;) { if (a < mn) mn = a; k++; a = t_swing_left(&T, a); if (a == c) break; }
      for (int a = c;;) { vert[a] = mn; a = t_swing_left(&T, a); if (a == c) break; }
      open[mn] = 0; if (ring) ring[mn] = k;
    } else {
      int k = 0; for (int a = l; a >= 0; a = t_swing_right(&T, a)) { vert[a] = l; k++; }
      open[l] = 1; if (ring) ring[l] = k + 1;
    }
    count++;
  }
  return count;
}

typedef struct { int32_t *p; int n, cap; } ivec;
static void iv_push(ivec *v, int32_t x) { if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = (int32_t *)realloc(v->p, 4 * (size_t)v->cap); } v->p[v->n++] = x; }
typedef struct { uint8_t *p; size_t n, cap; } bvec;
static void bv_push(bvec *v, uint8_t x) { if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = (uint8_t *)realloc(v->p, v->cap); } v->p[v->n++] = x; }

static inline uint32_t sym_of(int32_t v) { return v >= 0 ? ((uint32_t)v << 1) : ((((uint32_t)(-(v + 1))) << 1) | 1); }

/* wrap transform, encoder side (PredictionSchemeWrapEncodingTransform) */
typedef struct { int32_t lo, hi, max_dif, max_corr, min_corr; } wrapt;
static void wrap_init(wrapt *w, const int32_t *vals, size_t n) {
  int32_t lo = 0, hi = 0;
  if (n) { lo = hi = vals[0]; for (size_t i = 1; i < n; i++) { if (vals[i] < lo) lo = vals[i]; if (vals[i] > hi) hi = vals[i]; } }
  w->lo = lo; w->hi = hi; w->max_dif = 1 + hi - lo; w->max_corr = w->max_dif / 2; w->min_corr = -w->max_corr;
  if ((w->max_dif & 1) == 0) w->max_corr -= 1;
}
static inline int32_t wrap_corr(const wrapt *w, int32_t orig, int64_t pred) {
  int32_t p = pred < w->lo ? w->lo : (pred > w->hi ? w->hi : (int32_t)pred);
  int32_t c = orig - p;
  if (c < w->min_corr) c += w->max_dif; else if (c > w->max_corr) c -= w->max_dif;
  return c;
}

/* quantised octahedral coordinates of a float normal (AttributeOctahedronTransform) */
static void float_to_oct(const octb *t, const float *v, int32_t *s, int32_t *tt) {
  double abs_sum = fabs((double)v[0]) + fabs((double)v[1]) + fabs((double)v[2]);
  double sv[3];
  if (abs_sum > 1e-6) { double sc = 1.0 / abs_sum; sv[0] = v[0] * sc; sv[1] = v[1] * sc; sv[2] = v[2] * sc; }
  else { sv[0] = 1; sv[1] = 0; sv[2] = 0; }
  int32_t iv[3];
  iv[0] = (int32_t)floor(sv[0] * t->CEN + 0.5);
  iv[1] = (int32_t)floor(sv[1] * t->CEN + 0.5);
  iv[2] = t->CEN - abs(iv[0]) - abs(iv[1]);
  if (iv[2] < 0) { if (iv[1] > 0) iv[1] += iv[2]; else iv[1] -= iv[2]; iv[2] = 0; }
  if (sv[2] < 0) iv[2] *= -1;
  oct_vec_to_oct(t, iv, s, tt);
}
/* NormalOctahedronCanonicalizedEncodingTransform::ComputeCorrection (before ModMax/MakePositive of the caller) */
static void oct_corr(const octb *t, const int32_t orig_[2], const int32_t pred_[2], int32_t corr[2]) {
  int32_t os = orig_[0] - t->CEN, ot = orig_[1] - t->CEN, ps = pred_[0] - t->CEN, pt = pred_[1] - t->CEN;
  if ((abs(ps) + abs(pt)) > t->CEN) { oct_invert_diamond(t, &os, &ot); oct_invert_diamond(t, &ps, &pt); }
  int bl = (ps == 0 && pt == 0) || (ps < 0 && pt <= 0);
  if (!bl) { int rc = oct_rot_count(ps, pt); oct_rot(&os, &ot, rc); oct_rot(&ps, &pt, rc); }
  corr[0] = os - ps; corr[1] = ot - pt;
  if (corr[0] < 0) corr[0] += t->MAXQ;
  if (corr[1] < 0) corr[1] += t->MAXQ;
}

/* ---------------- sequential connectivity (MESH_SEQUENTIAL_ENCODING, what stock `draco_encoder -cl 0` selects) ----------------
 * Restated from the published bitstream description (MeshSequentialEncoder / SequentialAttributeEncodersController): no reference
 * fixture uses it, so it is pinned only by the round trip through drc_dec.c.  Points = the distinct (position, uv, normal) value
 * triples in order of first appearance over the corners (every face is kept, also degenerate ones); connectivity = the point index
 * of every corner, stored raw in the smallest type; ONE attribute decoder holds all attributes, each coded per point in point order
 * with the DIFFERENCE predictor (previous point's value; zeros for the first) through the wrap / canonicalised-octahedron
 * transform and the RAW rANS scheme; the de-quantisation parameters of all attributes follow the values of all attributes. */
static int drc_encode_sequential(const drc_enc_input *in, const drc_enc_params *prm, orc_buf *out) {
  const int qp = prm->qp, qt = prm->qt, qn = prm->qn;
  const int has_uv = in->uv && in->n_uv && in->idx_uv, has_nrm = in->nrm && in->n_nrm && in->idx_nrm;
  const uint32_t nf = in->nf, nc = 3 * nf;
  if (nf == 0) return -3;
  uint32_t *canon_p = (uint32_t *)malloc(4 * (size_t)(in->n_pos + 1)), *canon_u = NULL, *canon_n = NULL;
  dedup_values(in->pos, in->n_pos, 12, canon_p);
  if (has_uv) { canon_u = (uint32_t *)malloc(4 * (size_t)(in->n_uv + 1)); dedup_values(in->uv, in->n_uv, 8, canon_u); }
  if (has_nrm) { canon_n = (uint32_t *)malloc(4 * (size_t)(in->n_nrm + 1)); dedup_values(in->nrm, in->n_nrm, 12, canon_n); }
  /* points: first corner of every distinct (p, u, n) triple, two 64-bit map levels */
  int32_t *pu = (int32_t *)malloc(4 * (size_t)nc), *pid = (int32_t *)malloc(4 * (size_t)nc), *first = (int32_t *)malloc(4 * (size_t)nc);
  { emap E; emap_init(&E, nc);
    for (uint32_t c = 0; c < nc; c++) emap_put_min(&E, ((uint64_t)canon_p[in->idx_pos[c]] << 32) | (has_uv ? canon_u[in->idx_uv[c]] : 0u), (int32_t)c);
    for (uint32_t c = 0; c < nc; c++) pu[c] = emap_get(&E, ((uint64_t)canon_p[in->idx_pos[c]] << 32) | (has_uv ? canon_u[in->idx_uv[c]] : 0u));
    free(E.key); free(E.val);
    emap_init(&E, nc);
    for (uint32_t c = 0; c < nc; c++) emap_put_min(&E, ((uint64_t)(uint32_t)pu[c] << 32) | (has_nrm ? canon_n[in->idx_nrm[c]] : 0u), (int32_t)c);
    for (uint32_t c = 0; c < nc; c++) first[c] = emap_get(&E, ((uint64_t)(uint32_t)pu[c] << 32) | (has_nrm ? canon_n[in->idx_nrm[c]] : 0u));
    free(E.key); free(E.val); }
  uint32_t np = 0; int32_t *corner_of_point = (int32_t *)malloc(4 * (size_t)nc);
  for (uint32_t c = 0; c < nc; c++) if (first[c] == (int32_t)c) { corner_of_point[np] = (int32_t)c; pid[c] = (int32_t)np++; }
  for (uint32_t c = 0; c < nc; c++) pid[c] = pid[first[c]];
  ob_bytes(out, "DRACO", 5); ob_u8(out, 2); ob_u8(out, 2); ob_u8(out, 1); ob_u8(out, 0); ob_u16(out, 0);
  if (prm->method == 3) {                                                      /* connectivity_method 0 (compress_connectivity): signed differences to the previous index, as symbols */
    ob_varint(out, nf); ob_varint(out, np); ob_u8(out, 0);
    uint32_t *sy = (uint32_t *)malloc(4 * (size_t)nc + 4); int32_t last = 0;
    for (uint32_t c = 0; c < nc; c++) { const int32_t d = pid[c] - last; last = pid[c]; sy[c] = d < 0 ? (((uint32_t)(-d)) << 1) | 1u : ((uint32_t)d) << 1; }
    orc_encode_symbols(sy, nc, out); free(sy);
  } else {
  ob_varint(out, nf); ob_varint(out, np); ob_u8(out, 1);                       /* connectivity_method 1: indices stored directly */
  for (uint32_t c = 0; c < nc; c++) {
    const uint32_t v = (uint32_t)pid[c];
    if (np < 256) ob_u8(out, (uint8_t)v); else if (np < (1u << 16)) ob_u16(out, (uint16_t)v); else if (np < (1u << 21)) ob_varint(out, v); else ob_i32(out, (int32_t)v);
  }
  }
  const int natt = 1 + has_uv + has_nrm;
  ob_u8(out, 1);                                                               /* one attributes decoder */
  ob_varint(out, (uint64_t)natt);
  ob_u8(out, 0); ob_u8(out, 9); ob_u8(out, 3); ob_u8(out, 0); ob_varint(out, 0);
  { int id = 1;
    if (has_uv) { ob_u8(out, 3); ob_u8(out, 9); ob_u8(out, 2); ob_u8(out, 0); ob_varint(out, (uint64_t)id++); }
    if (has_nrm) { ob_u8(out, 1); ob_u8(out, 9); ob_u8(out, 3); ob_u8(out, 0); ob_varint(out, (uint64_t)id++); } }
  ob_u8(out, 2); if (has_uv) ob_u8(out, 2); if (has_nrm) ob_u8(out, 3);        /* sequential encoder types: quantization, quantization, normals */
  /* ---- portable values of every attribute ---- */
  float pmin[3], prange, umin[2] = {0, 0}, urange = 1.f;
  for (int a = 0; a < 2; a++) {
    if (a == 1 && !has_uv) continue;
    const int ncomp = a == 0 ? 3 : 2, q = a == 0 ? qp : qt;
    const float *src = a == 0 ? in->pos : in->uv; const uint32_t nval = a == 0 ? in->n_pos : in->n_uv;
    const uint32_t *canon = a == 0 ? canon_p : canon_u, *idx = a == 0 ? in->idx_pos : in->idx_uv;
    float mn[3], mx[3], range;
    for (int k = 0; k < ncomp; k++) { mn[k] = src[k]; mx[k] = src[k]; }
    for (uint32_t i = 1; i < nval; i++) for (int k = 0; k < ncomp; k++) { const float v = src[(size_t)ncomp * i + k]; if (v < mn[k]) mn[k] = v; if (v > mx[k]) mx[k] = v; }
    range = mx[0] - mn[0]; for (int k = 1; k < ncomp; k++) { const float d = mx[k] - mn[k]; if (d > range) range = d; }
    if (range == 0.f) range = 1.f;
    if (a == 0) { memcpy(pmin, mn, 12); prange = range; } else { memcpy(umin, mn, 8); urange = range; }
    const float inv = (float)((1u << q) - 1) / range;
    int32_t *Q = (int32_t *)malloc(4 * (size_t)ncomp * np + 4);
    for (uint32_t p = 0; p < np; p++) { const float *v = src + (size_t)ncomp * canon[idx[corner_of_point[p]]]; for (int k = 0; k < ncomp; k++) { float t = v[k] - mn[k]; t = t * inv; Q[(size_t)ncomp * p + k] = (int32_t)floorf(t + 0.5f); } }
    wrapt W; wrap_init(&W, Q, (size_t)ncomp * np);
    uint32_t *syms = (uint32_t *)malloc(4 * (size_t)ncomp * np + 4);
    for (uint32_t p = 0; p < np; p++) for (int k = 0; k < ncomp; k++) syms[(size_t)ncomp * p + k] = sym_of(wrap_corr(&W, Q[(size_t)ncomp * p + k], p ? (int64_t)Q[(size_t)ncomp * (p - 1) + k] : 0));
    ob_u8(out, 0); ob_u8(out, 1); ob_u8(out, 1);                               /* PREDICTION_DIFFERENCE, wrap transform, compressed */
    orc_encode_symbols(syms, (uint32_t)ncomp * np, out);
    ob_i32(out, W.lo); ob_i32(out, W.hi);
    free(Q); free(syms);
  }
  if (has_nrm) {
    octb ot; oct_init(&ot, qn);
    int32_t *O = (int32_t *)malloc(8 * (size_t)np + 8); uint32_t *syms = (uint32_t *)malloc(8 * (size_t)np + 8);
    for (uint32_t p = 0; p < np; p++) float_to_oct(&ot, in->nrm + 3 * (size_t)canon_n[in->idx_nrm[corner_of_point[p]]], &O[2 * p], &O[2 * p + 1]);
    for (uint32_t p = 0; p < np; p++) {
      const int32_t zero[2] = {0, 0}; int32_t corr[2];
      oct_corr(&ot, O + 2 * (size_t)p, p ? O + 2 * (size_t)(p - 1) : zero, corr);
      syms[2 * p] = (uint32_t)corr[0]; syms[2 * p + 1] = (uint32_t)corr[1];
    }
    ob_u8(out, 0); ob_u8(out, 3); ob_u8(out, 1);                               /* PREDICTION_DIFFERENCE, canonicalised octahedron transform */
    orc_encode_symbols(syms, 2 * np, out);
    ob_i32(out, ot.MAXQ); ob_i32(out, ot.CEN);
    free(O); free(syms);
  }
  /* ---- data needed by the portable transforms, attribute by attribute ---- */
  for (int k = 0; k < 3; k++) ob_f32(out, pmin[k]);
  ob_f32(out, prange); ob_u8(out, (uint8_t)qp);
  if (has_uv) { ob_f32(out, umin[0]); ob_f32(out, umin[1]); ob_f32(out, urange); ob_u8(out, (uint8_t)qt); }
  if (has_nrm) ob_u8(out, (uint8_t)qn);
  free(canon_p); free(canon_u); free(canon_n); free(pu); free(pid); free(first); free(corner_of_point);
  return 0;
}

int drc_encode(const drc_enc_input *in, const drc_enc_params *prm, orc_buf *out) {
  const int qp = prm->qp, qt = prm->qt, qn = prm->qn;
  if (qp < 1 || qp > 16 || qt < 1 || qt > 16 || qn < 2 || qn > 16) return -1;
  const int has_uv = in->uv && in->n_uv && in->idx_uv, has_nrm = in->nrm && in->n_nrm && in->idx_nrm;
  uint32_t nf_in = in->nf;
  for (uint32_t c = 0; c < 3 * nf_in; c++) {
    if (in->idx_pos[c] >= in->n_pos) return -2;
    if (has_uv && in->idx_uv[c] >= in->n_uv) return -2;
    if (has_nrm && in->idx_nrm[c] >= in->n_nrm) return -2;
  }
  if (prm->method == 2 || prm->method == 3) return drc_encode_sequential(in, prm, out);   /* 3: sequential with compressed indices (test streams for the decoders) */
  /* K2: value dedup (bitwise) */
  uint32_t *canon_p = (uint32_t *)malloc(4 * (size_t)(in->n_pos + 1)), *canon_u = NULL, *canon_n = NULL;
  dedup_values(in->pos, in->n_pos, 12, canon_p);
  if (has_uv) { canon_u = (uint32_t *)malloc(4 * (size_t)(in->n_uv + 1)); dedup_values(in->uv, in->n_uv, 8, canon_u); }
  if (has_nrm) { canon_n = (uint32_t *)malloc(4 * (size_t)(in->n_nrm + 1)); dedup_values(in->nrm, in->n_nrm, 12, canon_n); }
  /* drop degenerate faces (order preserving) */
  int nf = 0;
  int32_t *cp = (int32_t *)malloc(4 * 3 * (size_t)(nf_in + 1)), *cu = (int32_t *)malloc(4 * 3 * (size_t)(nf_in + 1)), *cn = (int32_t *)malloc(4 * 3 * (size_t)(nf_in + 1));
  for (uint32_t f = 0; f < nf_in; f++) {
    uint32_t a = canon_p[in->idx_pos[3 * f]], b = canon_p[in->idx_pos[3 * f + 1]], c = canon_p[in->idx_pos[3 * f + 2]];
    if (a == b || b == c || a == c) continue;
    for (int k = 0; k < 3; k++) {
      cp[3 * nf + k] = (int32_t)canon_p[in->idx_pos[3 * f + k]];
      cu[3 * nf + k] = has_uv ? (int32_t)canon_u[in->idx_uv[3 * f + k]] : 0;
      cn[3 * nf + k] = has_nrm ? (int32_t)canon_n[in->idx_nrm[3 * f + k]] : 0;
    }
    nf++;
  }
  free(canon_p); free(canon_u); free(canon_n);
  if (nf == 0) { free(cp); free(cu); free(cn); return -3; }
  const int nc = 3 * nf;
  /* K3: opposite corners */
  int32_t *opp = (int32_t *)malloc(4 * (size_t)nc);
  { emap E; emap_init(&E, (uint32_t)nc);
    for (int c = 0; c < nc; c++) emap_put_min(&E, ((uint64_t)(uint32_t)cp[c_nxt(c)] << 32) | (uint32_t)cp[c_prv(c)], c);
    for (int c = 0; c < nc; c++) {
      uint64_t a = (uint32_t)cp[c_nxt(c)], b = (uint32_t)cp[c_prv(c)];
      int32_t self = emap_get(&E, (a << 32) | b), o = emap_get(&E, (b << 32) | a);
      opp[c] = (self == c && o >= 0) ? o : -1;
    }
    free(E.key); free(E.val); }
  /* base fans */
  int32_t *vert = (int32_t *)malloc(4 * (size_t)nc), *ring = (int32_t *)calloc((size_t)nc, 4); uint8_t *vopen = (uint8_t *)calloc((size_t)nc, 1);
  int nverts = compute_fans(nf, opp, NULL, vert, vopen, ring);

  /* ---------------- K4: valence edgebreaker traversal ---------------- */
  uint8_t *fvis = (uint8_t *)calloc((size_t)nf, 1), *vvis = (uint8_t *)calloc((size_t)nc, 1);
  int32_t *vval = (int32_t *)malloc(4 * (size_t)(nc + nf + 1)); memcpy(vval, ring, 4 * (size_t)nc); int nvval = nc;
  int32_t *c2vm = (int32_t *)malloc(4 * (size_t)nc); memcpy(c2vm, vert, 4 * (size_t)nc);
  int32_t *f2split = (int32_t *)malloc(4 * (size_t)nf); for (int i = 0; i < nf; i++) f2split[i] = -1;
  ivec proc = {0}, initc = {0}, stack = {0}, ev_src = {0}, ev_spl = {0}, ev_edge = {0};
  ivec ctxs[6]; memset(ctxs, 0, sizeof(ctxs));
  ivec symseq = {0};                               /* the symbols in encoding order (the standard traversal stores them as bits) */
  bvec start_bits = {0};
  int last_sym_id = -1, nsplit = 0, prev_symbol = -1;
  enum { T_C = 0, T_S = 1, T_L = 3, T_R = 5, T_E = 7 };
  static const int topo2id[8] = { 0, 1, 0, 2, 0, 3, 0, 4 };
#define SWR(c) (opp[c_prv(c)] < 0 ? -1 : c_prv(opp[c_prv(c)]))
#define ENCODE_SYMBOL(symbol, last_corner) do { \
    int nx_ = c_nxt(last_corner), pv_ = c_prv(last_corner); \
    int active_valence = vval[c2vm[nx_]]; \
    switch (symbol) { \
      case T_C: case T_S: \
        vval[c2vm[nx_]] -= 1; vval[c2vm[pv_]] -= 1; \
        if ((symbol) == T_S) { \
          int nleft = 0, a_ = opp[pv_]; \
          while (a_ >= 0) { if (fvis[a_ / 3]) break; nleft++; a_ = opp[c_nxt(a_)]; } \
          vval[c2vm[last_corner]] = nleft + 1; \
          int newv = nvval, nright = 0; a_ = opp[nx_]; \
          while (a_ >= 0) { if (fvis[a_ / 3]) break; nright++; c2vm[c_nxt(a_)] = newv; a_ = opp[c_prv(a_)]; } \
          vval[nvval++] = nright + 1; \
        } break; \
      case T_R: vval[c2vm[last_corner]] -= 1; vval[c2vm[nx_]] -= 1; vval[c2vm[pv_]] -= 2; break; \
      case T_L: vval[c2vm[last_corner]] -= 1; vval[c2vm[nx_]] -= 2; vval[c2vm[pv_]] -= 1; break; \
      case T_E: vval[c2vm[last_corner]] -= 2; vval[c2vm[nx_]] -= 2; vval[c2vm[pv_]] -= 2; break; \
    } \
    if (prev_symbol != -1) { int cv = active_valence < 2 ? 2 : (active_valence > 7 ? 7 : active_valence); iv_push(&ctxs[cv - 2], topo2id[prev_symbol]); } \
    iv_push(&symseq, (symbol)); \
    prev_symbol = (symbol); } while (0)
#define CHECK_SPLIT(src_edge, nb_face) do { int sid_ = f2split[nb_face]; if (sid_ != -1) { iv_push(&ev_src, last_sym_id); iv_push(&ev_spl, sid_); iv_push(&ev_edge, (src_edge)); } } while (0)

  for (int f0 = 0; f0 < nf; f0++) {
    if (fvis[f0]) continue;
    /* FindInitFaceConfiguration */
    int ci = 3 * f0, interior = 1, start_corner = ci;
    for (int i = 0; i < 3; i++) {
      if (opp[ci] < 0) { interior = 0; start_corner = ci; break; }
      if (vopen[vert[ci]]) { int rcn = ci; while (rcn >= 0) { ci = rcn; rcn = SWR(rcn); } interior = 0; start_corner = c_prv(ci); break; }
      ci = c_nxt(ci);
    }
    bv_push(&start_bits, (uint8_t)interior);
    int from;
    if (interior) {
      ci = 3 * f0;
      vvis[vert[ci]] = 1; vvis[vert[c_nxt(ci)]] = 1; vvis[vert[c_prv(ci)]] = 1;
      fvis[f0] = 1;
      iv_push(&initc, c_nxt(ci));
      from = opp[c_nxt(ci)];
      if (from < 0 || fvis[from / 3]) continue;
    } else from = start_corner;
    /* EncodeConnectivityFromCorner */
    stack.n = 0; iv_push(&stack, from);
    while (stack.n > 0) {
      int corner = stack.p[stack.n - 1];
      if (corner < 0 || fvis[corner / 3]) { stack.n--; continue; }
      for (;;) {
        last_sym_id++;
        int face = corner / 3; fvis[face] = 1;
        iv_push(&proc, corner);
        int v = vert[corner], on_b = vopen[v];
        if (!vvis[v]) {
          vvis[v] = 1;
          if (!on_b) { ENCODE_SYMBOL(T_C, corner); corner = opp[c_nxt(corner)]; continue; }
        }
        int rcn = opp[c_nxt(corner)], lcn = opp[c_prv(corner)];
        int rvis = rcn < 0 ? 1 : fvis[rcn / 3], lvis = lcn < 0 ? 1 : fvis[lcn / 3];
        if (rvis) {
          if (rcn >= 0) CHECK_SPLIT(1, rcn / 3);
          if (lvis) {
            if (lcn >= 0) CHECK_SPLIT(0, lcn / 3);
            ENCODE_SYMBOL(T_E, corner); stack.n--; break;
          } else { ENCODE_SYMBOL(T_R, corner); corner = lcn; }
        } else {
          if (lvis) { if (lcn >= 0) CHECK_SPLIT(0, lcn / 3); ENCODE_SYMBOL(T_L, corner); corner = rcn; }
          else {
            ENCODE_SYMBOL(T_S, corner); nsplit++;
            f2split[face] = last_sym_id;
            stack.p[stack.n - 1] = lcn; iv_push(&stack, rcn);
            break;
          }
        }
      }
    }
  }
  const int nsym = last_sym_id + 1;
  /* ---------------- renumber into decoder order ---------------- */
  int32_t *old_of_new = (int32_t *)malloc(4 * (size_t)nc), *new_of_old = (int32_t *)malloc(4 * (size_t)nc);
  { int f = 0;
    for (int i = proc.n - 1; i >= 0; i--, f++) { int c = proc.p[i]; old_of_new[3 * f] = c; old_of_new[3 * f + 1] = c_nxt(c); old_of_new[3 * f + 2] = c_prv(c); }
    for (int i = 0; i < initc.n; i++, f++) { int c = initc.p[i]; old_of_new[3 * f] = c; old_of_new[3 * f + 1] = c_nxt(c); old_of_new[3 * f + 2] = c_prv(c); }
    if (f != nf) { fprintf(stderr, "drc_encode: traversal covered %d of %d faces\n", f, nf); return -10; } }
  for (int c = 0; c < nc; c++) new_of_old[old_of_new[c]] = c;
  int32_t *nopp = (int32_t *)malloc(4 * (size_t)nc), *npid = (int32_t *)malloc(4 * (size_t)nc), *nuid = (int32_t *)malloc(4 * (size_t)nc), *nnid = (int32_t *)malloc(4 * (size_t)nc);
  for (int c = 0; c < nc; c++) { int o = old_of_new[c]; nopp[c] = opp[o] < 0 ? -1 : new_of_old[opp[o]]; npid[c] = cp[o]; nuid[c] = cu[o]; nnid[c] = cn[o]; }
  int32_t *bvert = (int32_t *)malloc(4 * (size_t)nc), *ident = (int32_t *)malloc(4 * (size_t)nc); uint8_t *bopen = (uint8_t *)calloc((size_t)nc, 1);
  compute_fans(nf, nopp, NULL, bvert, bopen, NULL);
  for (int c = 0; c < nc; c++) ident[c] = c;

  /* ---------------- attribute seams ---------------- */
  const int nad = has_uv + has_nrm;
  const int32_t *att_ids[2]; int att_kind[2];   /* kind: 0 uv, 1 normal */
  { int k = 0; if (has_uv) { att_ids[k] = nuid; att_kind[k++] = 0; } if (has_nrm) { att_ids[k] = nnid; att_kind[k++] = 1; } }
  uint8_t *seam[2] = {0}; int interior_seams[2] = {0, 0}; bvec seam_bits[2]; memset(seam_bits, 0, sizeof(seam_bits));
  for (int i = 0; i < nad; i++) {
    seam[i] = (uint8_t *)calloc((size_t)nc, 1);
    for (int c = 0; c < nc; c++) {
      int oc = nopp[c];
      if (oc < 0) { seam[i][c] = 1; continue; }
      if (att_ids[i][c_nxt(c)] != att_ids[i][c_prv(oc)] || att_ids[i][c_prv(c)] != att_ids[i][c_nxt(oc)]) { seam[i][c] = 1; interior_seams[i] = 1; }
    }
  }
  for (int f = 0; f < nf; f++) for (int k = 0; k < 3; k++) {
    int c = 3 * f + k, oc = nopp[c];
    if (oc < 0 || oc / 3 < f) continue;
    for (int i = 0; i < nad; i++) bv_push(&seam_bits[i], seam[i][c]);
  }

  /* ---------------- connectivity section ---------------- */
  ob_bytes(out, "DRACO", 5); ob_u8(out, 2); ob_u8(out, 2); ob_u8(out, 1); ob_u8(out, 1); ob_u16(out, 0);
  ob_u8(out, prm->method == 1 ? 0 : 2);            /* traversal: MESH_EDGEBREAKER_STANDARD_ENCODING (0) / _VALENCE_ENCODING (2) */
  ob_varint(out, (uint64_t)nverts); ob_varint(out, (uint64_t)nf); ob_u8(out, (uint8_t)nad);
  ob_varint(out, (uint64_t)nsym); ob_varint(out, (uint64_t)nsplit);
  ob_varint(out, (uint64_t)ev_src.n);
  { int last = 0;
    for (int i = 0; i < ev_src.n; i++) { ob_varint(out, (uint64_t)(ev_src.p[i] - last)); ob_varint(out, (uint64_t)(ev_src.p[i] - ev_spl.p[i])); last = ev_src.p[i]; }
    if (ev_src.n > 0) { int nb = (ev_src.n + 7) / 8; for (int j = 0; j < nb; j++) { uint8_t v = 0; for (int k = 0; k < 8 && 8 * j + k < ev_src.n; k++) v |= (uint8_t)((ev_edge.p[8 * j + k] & 1) << k); ob_u8(out, v); } } }
  if (prm->method == 1) {
    /* standard traversal (MeshEdgebreakerTraversalEncoder::Done): the symbols, last first, as bit patterns C = 0 (1 bit), S / L / R / E =
     * 1 / 3 / 5 / 7 (3 bits), LSB first, in a size-prefixed bit sequence; then the start-face and seam bit streams */
    size_t nbits = 0; for (int i = 0; i < symseq.n; i++) nbits += symseq.p[i] == 0 ? 1 : 3;
    const size_t nbytes = (nbits + 7) / 8; uint8_t *bits = (uint8_t *)calloc(nbytes + 1, 1); size_t bo = 0;
    for (int i = symseq.n - 1; i >= 0; i--) { const int sy = symseq.p[i], len = sy == 0 ? 1 : 3; for (int k = 0; k < len; k++, bo++) if ((sy >> k) & 1) bits[bo >> 3] |= (uint8_t)(1u << (bo & 7)); }
    ob_varint(out, (uint64_t)nbytes); ob_bytes(out, bits, nbytes); free(bits);
    orc_rabs_encode(start_bits.p, start_bits.n, out);
    for (int i = 0; i < nad; i++) orc_rabs_encode(seam_bits[i].p, seam_bits[i].n, out);
  } else {
    orc_rabs_encode(start_bits.p, start_bits.n, out);
    for (int i = 0; i < nad; i++) orc_rabs_encode(seam_bits[i].p, seam_bits[i].n, out);
    for (int i = 0; i < 6; i++) { ob_varint(out, (uint64_t)ctxs[i].n); if (ctxs[i].n > 0) orc_encode_symbols((const uint32_t *)ctxs[i].p, (uint32_t)ctxs[i].n, out); }
  }

  /* ---------------- attribute decoder headers ---------------- */
  const int ndec = 1 + nad;
  int dec_type[3] = {0, 0, 0};
  for (int i = 0; i < nad; i++) dec_type[1 + i] = interior_seams[i] ? 1 : 0;
  ob_u8(out, (uint8_t)ndec);
  ob_u8(out, 0xff); ob_u8(out, 0); ob_u8(out, 0);
  for (int i = 0; i < nad; i++) { ob_u8(out, (uint8_t)i); ob_u8(out, (uint8_t)dec_type[1 + i]); ob_u8(out, 0); }
  ob_varint(out, 1); ob_u8(out, 0); ob_u8(out, 9); ob_u8(out, 3); ob_u8(out, 0); ob_varint(out, 0); ob_u8(out, 2);
  for (int i = 0; i < nad; i++) {
    ob_varint(out, 1);
    if (att_kind[i] == 0) { ob_u8(out, 3); ob_u8(out, 9); ob_u8(out, 2); ob_u8(out, 0); ob_varint(out, (uint64_t)(1 + i)); ob_u8(out, 2); }
    else { ob_u8(out, 1); ob_u8(out, 9); ob_u8(out, 3); ob_u8(out, 0); ob_varint(out, (uint64_t)(1 + i)); ob_u8(out, 3); }
  }

  /* ---------------- base traversal + positions ---------------- */
  int32_t *b_order = (int32_t *)malloc(4 * (size_t)nc), *b_v2d = (int32_t *)malloc(4 * (size_t)nc);
  ctab B = { nf, nc, nopp, NULL, bvert, ident };
  int b_n = traverse(&B, b_order, b_v2d);
  if (b_n != nverts) { fprintf(stderr, "drc_encode: base traversal %d entries vs %d vertices\n", b_n, nverts); return -11; }
  int32_t *P = (int32_t *)malloc(4 * 3 * (size_t)b_n);
  float pmin[3], prange;
  { float mx[3];
    for (int k = 0; k < 3; k++) { pmin[k] = in->pos[k]; mx[k] = in->pos[k]; }
    for (uint32_t i = 1; i < in->n_pos; i++) for (int k = 0; k < 3; k++) { float v = in->pos[3 * i + k]; if (v < pmin[k]) pmin[k] = v; if (v > mx[k]) mx[k] = v; }
    prange = mx[0] - pmin[0]; for (int k = 1; k < 3; k++) { float d = mx[k] - pmin[k]; if (d > prange) prange = d; }
    if (prange == 0.f) prange = 1.f;
    float inv = (float)((1u << qp) - 1) / prange;
    for (int p = 0; p < b_n; p++) { const float *v = in->pos + 3 * (size_t)npid[b_order[p]]; for (int k = 0; k < 3; k++) { float t = (v[k] - pmin[k]); t = t * inv; P[3 * p + k] = (int32_t)floorf(t + 0.5f); } } }
  {
    wrapt W; wrap_init(&W, P, 3 * (size_t)b_n);
    uint32_t *syms = (uint32_t *)malloc(4 * 3 * (size_t)b_n + 4);
    for (int p = 0; p < b_n; p++) {
      int64_t pred[3] = {0, 0, 0}; int have = 0;
      if (p > 0) {
        int ci = b_order[p], oci = nopp[ci];
        if (oci >= 0) { int a = b_v2d[bvert[oci]], bn = b_v2d[bvert[c_nxt(oci)]], bp = b_v2d[bvert[c_prv(oci)]];
          if (a < p && bn < p && bp < p) { for (int k = 0; k < 3; k++) pred[k] = (int64_t)P[3 * bn + k] + P[3 * bp + k] - P[3 * a + k]; have = 1; } }
        if (!have) for (int k = 0; k < 3; k++) pred[k] = P[3 * (p - 1) + k];
      }
      for (int k = 0; k < 3; k++) syms[3 * p + k] = sym_of(wrap_corr(&W, P[3 * p + k], pred[k]));
    }
    ob_u8(out, 1); ob_u8(out, 1); ob_u8(out, 1);
    orc_encode_symbols(syms, 3 * (uint32_t)b_n, out);
    ob_i32(out, W.lo); ob_i32(out, W.hi);
    for (int k = 0; k < 3; k++) ob_f32(out, pmin[k]);
    ob_f32(out, prange); ob_u8(out, (uint8_t)qp);
    free(syms);
  }

  /* ---------------- non-position attributes ---------------- */
  for (int i = 0; i < nad; i++) {
    ctab X = B; const int32_t *order = b_order, *v2d = b_v2d; int ne = b_n;
    int32_t *avert = NULL, *a_order = NULL, *a_v2d = NULL; uint8_t *aopen = NULL;
    if (dec_type[1 + i] == 1) {
      avert = (int32_t *)malloc(4 * (size_t)nc); aopen = (uint8_t *)calloc((size_t)nc, 1);
      compute_fans(nf, nopp, seam[i], avert, aopen, NULL);
      X.edge_seam = seam[i]; X.c2v = avert;
      a_order = (int32_t *)malloc(4 * (size_t)nc); a_v2d = (int32_t *)malloc(4 * (size_t)nc);
      ne = traverse(&X, a_order, a_v2d); order = a_order; v2d = a_v2d;
    }
    if (att_kind[i] == 0) {
      /* ---- UV: quantise, tex-coord-portable prediction, wrap ---- */
      int32_t *U = (int32_t *)malloc(4 * 2 * (size_t)ne + 4);
      float umin[2], mx[2], urange;
      for (int k = 0; k < 2; k++) { umin[k] = in->uv[k]; mx[k] = in->uv[k]; }
      for (uint32_t j = 1; j < in->n_uv; j++) for (int k = 0; k < 2; k++) { float v = in->uv[2 * j + k]; if (v < umin[k]) umin[k] = v; if (v > mx[k]) mx[k] = v; }
      urange = mx[0] - umin[0]; if (mx[1] - umin[1] > urange) urange = mx[1] - umin[1];
      if (urange == 0.f) urange = 1.f;
      float inv = (float)((1u << qt) - 1) / urange;
      for (int p = 0; p < ne; p++) { const float *v = in->uv + 2 * (size_t)nuid[order[p]]; for (int k = 0; k < 2; k++) { float t = v[k] - umin[k]; t = t * inv; U[2 * p + k] = (int32_t)floorf(t + 0.5f); } }
      wrapt W; wrap_init(&W, U, 2 * (size_t)ne);
      uint32_t *syms = (uint32_t *)malloc(4 * 2 * (size_t)ne + 4);
      bvec ori = {0};
      for (int p = ne - 1; p >= 0; p--) {
        int c = order[p], cnx = c_nxt(c), cpv = c_prv(c);
        int nd = v2d[X.c2v[cnx]], pd = v2d[X.c2v[cpv]];
        int64_t pred[2] = {0, 0}; int have = 0;
        if (pd < p && nd < p) {
          int64_t nuv[2] = { U[2 * nd], U[2 * nd + 1] }, puv[2] = { U[2 * pd], U[2 * pd + 1] };
          if (puv[0] == nuv[0] && puv[1] == nuv[1]) { pred[0] = puv[0]; pred[1] = puv[1]; have = 1; }
          else {
            const int32_t *tip = P + 3 * b_v2d[bvert[c]], *np_ = P + 3 * b_v2d[bvert[cnx]], *pp_ = P + 3 * b_v2d[bvert[cpv]];
            int64_t pn[3], pn2 = 0, dd = 0;
            for (int k = 0; k < 3; k++) { pn[k] = (int64_t)pp_[k] - np_[k]; pn2 += pn[k] * pn[k]; }
            if (pn2 != 0) {
              for (int k = 0; k < 3; k++) dd += pn[k] * ((int64_t)tip[k] - np_[k]);
              int64_t pnuv[2] = { puv[0] - nuv[0], puv[1] - nuv[1] };
              int64_t xuv[2] = { nuv[0] * pn2 + dd * pnuv[0], nuv[1] * pn2 + dd * pnuv[1] };
              int64_t cx2 = 0;
              for (int k = 0; k < 3; k++) { int64_t xp = np_[k] + (dd * pn[k]) / pn2; int64_t e = tip[k] - xp; cx2 += e * e; }
              int64_t ns_ = (int64_t)orc_isqrt((uint64_t)cx2 * (uint64_t)pn2);
              int64_t cxuv[2] = { pnuv[1] * ns_, -pnuv[0] * ns_ };
              int64_t p0[2] = { (xuv[0] + cxuv[0]) / pn2, (xuv[1] + cxuv[1]) / pn2 }, p1[2] = { (xuv[0] - cxuv[0]) / pn2, (xuv[1] - cxuv[1]) / pn2 };
              int64_t cu0 = U[2 * p], cu1 = U[2 * p + 1];
              int64_t d0 = (cu0 - p0[0]) * (cu0 - p0[0]) + (cu1 - p0[1]) * (cu1 - p0[1]);
              int64_t d1 = (cu0 - p1[0]) * (cu0 - p1[0]) + (cu1 - p1[1]) * (cu1 - p1[1]);
              if (d0 < d1) { pred[0] = p0[0]; pred[1] = p0[1]; bv_push(&ori, 1); } else { pred[0] = p1[0]; pred[1] = p1[1]; bv_push(&ori, 0); }
              have = 1;
            }
          }
        }
        if (!have) {
          if (nd < p) { pred[0] = U[2 * nd]; pred[1] = U[2 * nd + 1]; }
          else if (p > 0) { pred[0] = U[2 * (p - 1)]; pred[1] = U[2 * (p - 1) + 1]; }
        }
        /* predicted values are cast to int (32-bit) before the transform */
        for (int k = 0; k < 2; k++) syms[2 * p + k] = sym_of(wrap_corr(&W, U[2 * p + k], (int64_t)(int32_t)pred[k]));
      }
      ob_u8(out, 5); ob_u8(out, 1); ob_u8(out, 1);
      orc_encode_symbols(syms, 2 * (uint32_t)ne, out);
      ob_i32(out, (int32_t)ori.n);
      { uint8_t *bits = (uint8_t *)malloc(ori.n + 1); int last = 1; for (size_t k = 0; k < ori.n; k++) { bits[k] = (uint8_t)(ori.p[k] == last); last = ori.p[k]; }
        orc_rabs_encode(bits, ori.n, out); free(bits); }
      ob_i32(out, W.lo); ob_i32(out, W.hi);
      ob_f32(out, umin[0]); ob_f32(out, umin[1]); ob_f32(out, urange); ob_u8(out, (uint8_t)qt);
      free(U); free(syms); free(ori.p);
    } else {
      /* ---- normals: octahedral quantisation, geometric-normal prediction, canonicalised transform ---- */
      octb ot; oct_init(&ot, qn);
      int32_t *O = (int32_t *)malloc(4 * 2 * (size_t)ne + 4);
      for (int p = 0; p < ne; p++) float_to_oct(&ot, in->nrm + 3 * (size_t)nnid[order[p]], &O[2 * p], &O[2 * p + 1]);
      uint32_t *syms = (uint32_t *)malloc(4 * 2 * (size_t)ne + 4);
      uint8_t *flips = (uint8_t *)malloc((size_t)ne + 1);
      for (int d = 0; d < ne; d++) {
        int c0 = order[d];
        const int32_t *cenp = P + 3 * b_v2d[bvert[c0]];
        int64_t N[3] = {0, 0, 0};
        int c = c0, left = 1;
        while (c >= 0) {
          const int32_t *a = P + 3 * b_v2d[bvert[c_nxt(c)]], *bb = P + 3 * b_v2d[bvert[c_prv(c)]];
          int64_t dn[3], dp[3];
          for (int k = 0; k < 3; k++) { dn[k] = (int64_t)a[k] - cenp[k]; dp[k] = (int64_t)bb[k] - cenp[k]; }
          N[0] += dn[1] * dp[2] - dn[2] * dp[1]; N[1] += dn[2] * dp[0] - dn[0] * dp[2]; N[2] += dn[0] * dp[1] - dn[1] * dp[0];
          if (left) { c = t_swing_left(&X, c); if (c == c0) break; if (c < 0) { left = 0; c = t_swing_right(&X, c0); } }
          else c = t_swing_right(&X, c);
        }
        int64_t s = llabs(N[0]) + llabs(N[1]) + llabs(N[2]);
        if (s > (1 << 29)) { int64_t qd = s / (1 << 29); for (int k = 0; k < 3; k++) N[k] /= qd; }
        int32_t pv[3], ppos[2], pneg[2], cpos[2], cneg[2];
        oct_canon_vec(&ot, N, pv);
        oct_vec_to_oct(&ot, pv, &ppos[0], &ppos[1]);
        pv[0] = -pv[0]; pv[1] = -pv[1]; pv[2] = -pv[2];
        oct_vec_to_oct(&ot, pv, &pneg[0], &pneg[1]);
        oct_corr(&ot, O + 2 * d, ppos, cpos); oct_corr(&ot, O + 2 * d, pneg, cneg);
        for (int k = 0; k < 2; k++) { cpos[k] = oct_modmax(&ot, cpos[k]); cneg[k] = oct_modmax(&ot, cneg[k]); }
        const int32_t *ch;
        if (abs(cpos[0]) + abs(cpos[1]) < abs(cneg[0]) + abs(cneg[1])) { flips[d] = 0; ch = cpos; } else { flips[d] = 1; ch = cneg; }
        for (int k = 0; k < 2; k++) syms[2 * d + k] = (uint32_t)(ch[k] < 0 ? ch[k] + ot.MAXQ : ch[k]);
      }
      ob_u8(out, 6); ob_u8(out, 3); ob_u8(out, 1);
      orc_encode_symbols(syms, 2 * (uint32_t)ne, out);
      ob_i32(out, ot.MAXQ); ob_i32(out, ot.CEN);
      orc_rabs_encode(flips, (size_t)ne, out);
      ob_u8(out, (uint8_t)qn);
      free(O); free(syms); free(flips);
    }
    free(avert); free(aopen); free(a_order); free(a_v2d);
  }

  free(cp); free(cu); free(cn); free(opp); free(vert); free(ring); free(vopen); free(fvis); free(vvis); free(vval); free(c2vm); free(f2split);
  free(proc.p); free(initc.p); free(stack.p); free(ev_src.p); free(ev_spl.p); free(ev_edge.p); for (int i = 0; i < 6; i++) free(ctxs[i].p);
  free(start_bits.p); free(old_of_new); free(new_of_old); free(nopp); free(npid); free(nuid); free(nnid); free(bvert); free(ident); free(bopen);
  for (int i = 0; i < nad; i++) { free(seam[i]); free(seam_bits[i].p); }
  free(b_order); free(b_v2d); free(P);
  return 0;
}
