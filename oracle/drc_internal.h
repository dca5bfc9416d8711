/* oracle/drc_internal.h — TEST INFRASTRUCTURE (see oracle_common.h).
 * Pieces shared by the oracle's Draco decoder and encoder restatement: corner-table view,
 * DepthFirstTraverser, wrap transform (A.6), octahedral toolbox (A.9/D.6).
 */
#ifndef UVOL_DRC_INTERNAL_H
#define UVOL_DRC_INTERNAL_H
#include "drc_oracle.h"
#include <math.h>
#if defined(__GNUC__)
#define ORC_UNUSED __attribute__((unused))
#else
#define ORC_UNUSED
#endif

/* corner table view: base table, or attribute table (seam-masked opposite + own vertex ids) */
typedef struct {
  int nf, nverts;
  const int32_t *opp_base;
  const uint8_t *edge_seam;   /* NULL => base table */
  const int32_t *c2v, *lm;
} ctab;
ORC_UNUSED static inline int t_opp(const ctab *t, int c) {
  if (c < 0) return ORC_INV;
  if (t->edge_seam && t->edge_seam[c]) return ORC_INV;
  return t->opp_base[c];
}
ORC_UNUSED static inline int t_swing_left(const ctab *t, int c) { int o = t_opp(t, c_nxt(c)); return o < 0 ? ORC_INV : c_nxt(o); }
ORC_UNUSED static inline int t_swing_right(const ctab *t, int c) { int o = t_opp(t, c_prv(c)); return o < 0 ? ORC_INV : c_prv(o); }

/* DepthFirstTraverser. order = data_to_corner, v2d = vertex_to_data */
ORC_UNUSED static int traverse(const ctab *t, int32_t *order, int32_t *v2d) {
  int nf = t->nf, n = 0, sp;
  uint8_t *fv = (uint8_t *)calloc(nf ? nf : 1, 1), *vv = (uint8_t *)calloc(t->nverts ? t->nverts : 1, 1);
  int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nf + 8));
  for (int i = 0; i < t->nverts; i++) v2d[i] = -1;
#define VISIT(v, c) do { vv[v] = 1; v2d[v] = n; order[n++] = (c); } while (0)
#define FVIS(c) ((c) < 0 ? 1 : fv[(c) / 3])
  for (int f = 0; f < nf; f++) {
    int cid = 3 * f;
    if (fv[f]) continue;
    sp = 0; stack[sp++] = cid;
    { int cn = c_nxt(cid), cp = c_prv(cid);
      if (!vv[t->c2v[cn]]) VISIT(t->c2v[cn], cn);
      if (!vv[t->c2v[cp]]) VISIT(t->c2v[cp], cp); }
    while (sp > 0) {
      cid = stack[sp - 1];
      if (cid < 0 || fv[cid / 3]) { sp--; continue; }
      for (;;) {
        fv[cid / 3] = 1;
        int v = t->c2v[cid];
        int lmc = t->lm[v];
        int ob = (lmc < 0) || (t_swing_left(t, lmc) < 0);
        if (!vv[v]) {
          VISIT(v, cid);
          if (!ob) { cid = t_opp(t, c_nxt(cid)); continue; }
        }
        int rc = t_opp(t, c_nxt(cid)), lc = t_opp(t, c_prv(cid));
        if (FVIS(rc)) {
          if (FVIS(lc)) { sp--; break; }
          cid = lc;
        } else {
          if (FVIS(lc)) cid = rc;
          else { stack[sp - 1] = lc; stack[sp++] = rc; break; }
        }
      }
    }
  }
#undef VISIT
#undef FVIS
  free(fv); free(vv); free(stack);
  return n;
}

ORC_UNUSED static inline int32_t sgn_sym(uint32_t s) { return (s & 1) ? -(int32_t)(s >> 1) - 1 : (int32_t)(s >> 1); }
ORC_UNUSED static inline int32_t wrap_orig(int32_t pred, int32_t corr, int32_t lo, int32_t hi) {
  int32_t md = 1 + hi - lo;
  int32_t v = (pred < lo ? lo : (pred > hi ? hi : pred)) + corr;
  if (v > hi) v -= md; else if (v < lo) v += md;
  return v;
}

/* ---- octahedral toolbox, q bits ---- */
typedef struct { int q, MAXQ, MAXV, CEN; } octb;
ORC_UNUSED static void oct_init(octb *t, int q) { t->q = q; t->MAXQ = (1 << q) - 1; t->MAXV = t->MAXQ - 1; t->CEN = t->MAXV / 2; }
ORC_UNUSED static void oct_canon_vec(const octb *t, const int64_t v[3], int32_t o[3]) {
  int64_t s = llabs(v[0]) + llabs(v[1]) + llabs(v[2]);
  if (s == 0) { o[0] = t->CEN; o[1] = 0; o[2] = 0; return; }
  int64_t a = (v[0] * t->CEN) / s, b = (v[1] * t->CEN) / s;
  int64_t c = t->CEN - llabs(a) - llabs(b);
  if (v[2] < 0) c = -c;
  o[0] = (int32_t)a; o[1] = (int32_t)b; o[2] = (int32_t)c;
}
ORC_UNUSED static void oct_canon_oct(const octb *t, int32_t *s, int32_t *tt) {
  int S = *s, T = *tt, MAXV = t->MAXV, CEN = t->CEN;
  if ((S == 0 && T == 0) || (S == 0 && T == MAXV) || (S == MAXV && T == 0)) { *s = MAXV; *tt = MAXV; return; }
  if (S == 0 && T > CEN) T = CEN - (T - CEN);
  else if (S == MAXV && T < CEN) T = CEN + (CEN - T);
  else if (T == MAXV && S < CEN) S = CEN + (CEN - S);
  else if (T == 0 && S > CEN) S = CEN - (S - CEN);
  *s = S; *tt = T;
}
ORC_UNUSED static void oct_vec_to_oct(const octb *t, const int32_t v[3], int32_t *s, int32_t *tt) {
  if (v[0] >= 0) { *s = v[1] + t->CEN; *tt = v[2] + t->CEN; }
  else {
    *s = v[1] < 0 ? abs(v[2]) : t->MAXV - abs(v[2]);
    *tt = v[2] < 0 ? abs(v[1]) : t->MAXV - abs(v[1]);
  }
  oct_canon_oct(t, s, tt);
}
ORC_UNUSED static void oct_invert_diamond(const octb *t, int32_t *s, int32_t *tt) {
  int32_t S = *s, T = *tt, ss, st;
  if (S >= 0 && T >= 0) { ss = 1; st = 1; }
  else if (S <= 0 && T <= 0) { ss = -1; st = -1; }
  else { ss = S > 0 ? 1 : -1; st = T > 0 ? 1 : -1; }
  int32_t cs = ss * t->CEN, ct = st * t->CEN;
  int32_t us = 2 * S - cs, ut = 2 * T - ct;
  if (ss * st >= 0) { int32_t tmp = us; us = -ut; ut = -tmp; }
  else { int32_t tmp = us; us = ut; ut = tmp; }
  us += cs; ut += ct;
  *s = us / 2; *tt = ut / 2;
}
ORC_UNUSED static int oct_rot_count(int32_t x, int32_t y) {
  if (x == 0) return y == 0 ? 0 : (y > 0 ? 3 : 1);
  if (x > 0) return y >= 0 ? 2 : 1;
  return y <= 0 ? 0 : 3;
}
ORC_UNUSED static void oct_rot(int32_t *x, int32_t *y, int c) {
  int32_t X = *x, Y = *y;
  if (c == 1) { *x = Y; *y = -X; } else if (c == 2) { *x = -X; *y = -Y; } else if (c == 3) { *x = -Y; *y = X; }
}
ORC_UNUSED static int32_t oct_modmax(const octb *t, int32_t x) { if (x > t->CEN) return x - t->MAXQ; if (x < -t->CEN) return x + t->MAXQ; return x; }
ORC_UNUSED static void oct_orig_value(const octb *t, const int32_t pred[2], const int32_t corr[2], int32_t out[2]) {
  int32_t ps = pred[0] - t->CEN, pt = pred[1] - t->CEN;
  int ind = (abs(ps) + abs(pt)) <= t->CEN;
  if (!ind) oct_invert_diamond(t, &ps, &pt);
  int bl = (ps == 0 && pt == 0) || (ps < 0 && pt <= 0);
  int rc = oct_rot_count(ps, pt);
  if (!bl) oct_rot(&ps, &pt, rc);
  int32_t os = oct_modmax(t, ps + corr[0]), ot = oct_modmax(t, pt + corr[1]);
  if (!bl) oct_rot(&os, &ot, (4 - rc) % 4);
  if (!ind) oct_invert_diamond(t, &os, &ot);
  out[0] = os + t->CEN; out[1] = ot + t->CEN;
}


#endif
