/* oracle/drc_dec.c — TEST INFRASTRUCTURE (see oracle_common.h).
 * Draco 2.2 triangular-mesh decoder (edgebreaker, valence traversal) — the conformance pin of the
 * whole geometry path: reference fixtures pin this decoder, this decoder pins every encoder output.
 * Follows SURVEY.md A.1–A.9 and the executed listings D.2–D.6.  Consumer it stands in for:
 * src/lib/DRACOLoader.js:470-590 (draco_decoder 1.4.3 WASM, src/V2/player.ts:101).
 */
#include "drc_internal.h"
#include <stdio.h>

typedef struct { const uint8_t *b; size_t n, o; int err; } rdr;
static uint8_t r_u8(rdr *r) { if (r->o + 1 > r->n) { r->err = 1; return 0; } return r->b[r->o++]; }
static int32_t r_i32(rdr *r) { int32_t v = 0; if (r->o + 4 > r->n) { r->err = 1; return 0; } memcpy(&v, r->b + r->o, 4); r->o += 4; return v; }
static float r_f32(rdr *r) { float v = 0; if (r->o + 4 > r->n) { r->err = 1; return 0; } memcpy(&v, r->b + r->o, 4); r->o += 4; return v; }
static uint32_t r_varint(rdr *r) {
  uint64_t v = 0; int s = 0;
  for (;;) { if (r->o >= r->n || s > 35) { r->err = 1; return 0; } uint8_t c = r->b[r->o++]; v |= (uint64_t)(c & 0x7f) << s; s += 7; if (c < 0x80) break; }
  return (uint32_t)v;
}

void drc_mesh_free(drc_mesh *m) {
  free(m->opp); free(m->c2v);
  for (int i = 0; i < m->natt; i++) { free(m->att[i].vals); free(m->att[i].corner_to_entry); }
  memset(m, 0, sizeof(*m));
}


/* wrap / octahedron helpers of the attribute loops below are in drc_internal.h */

/* Sequential connectivity (encoder_method 0, MeshSequentialDecoder): restated from the published bitstream description, exercised
 * by no reference fixture; pinned by the round trip with drc_enc.c (method 2).  One attributes decoder, every attribute has one
 * entry per point, decoded in point order with the DIFFERENCE predictor (or none). */
static int drc_decode_sequential(const uint8_t *b, size_t n, drc_mesh *m) {
  rdr R = { b, n, 11, 0 }, *r = &R;
  int rc = 0;
  const int nf = (int)r_varint(r), np = (int)r_varint(r), cm = r_u8(r);
  if (r->err || nf <= 0 || np <= 0 || nf > (1 << 28) || np > (1 << 28)) return -6;
  m->method = 0; m->traversal = cm; m->nf = nf; m->npoints = np; m->nev = np;
  int32_t *faces = (int32_t *)malloc(4 * 3 * (size_t)nf);
  if (cm == 0) {                                       /* compressed indices: signed differences to the previous index, as symbols */
    uint32_t *sy = (uint32_t *)malloc(4 * 3 * (size_t)nf + 4);
    if (orc_decode_symbols_nc(b, n, &r->o, 3 * (uint32_t)nf, 1, sy, NULL)) rc = -8;
    int32_t last = 0;
    for (int i = 0; i < 3 * nf && !rc; i++) { int32_t d = (int32_t)(sy[i] >> 1); if (sy[i] & 1) d = -d; last += d; faces[i] = last; }
    free(sy);
  } else if (cm == 1) {
    for (int i = 0; i < 3 * nf; i++) {
      if (np < 256) faces[i] = r_u8(r);
      else if (np < (1 << 16)) { uint32_t lo = r_u8(r), hi = r_u8(r); faces[i] = (int32_t)(lo | (hi << 8)); }
      else if (np < (1 << 21)) faces[i] = (int32_t)r_varint(r);
      else faces[i] = r_i32(r);
    }
  } else rc = -5;
  if (r->err) rc = -8;
  for (int i = 0; i < 3 * nf && !rc; i++) if (faces[i] < 0 || faces[i] >= np) rc = -19;
  m->conn_end = r->o;
  if (rc) { free(faces); return rc; }
  const int ndec = r_u8(r); if (r->err || ndec != 1) { free(faces); return -20; }
  const int natt = (int)r_varint(r); if (r->err || natt < 1 || natt > 8) { free(faces); return -22; }
  for (int d = 0; d < natt; d++) { drc_att *A = &m->att[d]; A->att_type = r_u8(r); A->data_type = r_u8(r); A->ncomp = r_u8(r); (void)r_u8(r); A->unique_id = (int)r_varint(r); A->att_data_id = -1; A->dec_type = 0; }
  for (int d = 0; d < natt; d++) m->att[d].seq_type = r_u8(r);
  m->natt = natt; m->hdr_end = r->o;
  if (r->err) { free(faces); m->natt = 0; return -22; }
  for (int d = 0; d < natt && !rc; d++) {
    drc_att *A = &m->att[d];
    A->n = np; A->sec_begin = r->o;
    A->pred_method = (int8_t)r_u8(r);
    if (A->pred_method != -2) A->transform = (int8_t)r_u8(r);
    const int compressed = r_u8(r);
    const int nc = (A->seq_type == 3) ? 2 : A->ncomp; A->ncomp_port = nc;
    if (nc < 1 || nc > 4 || (A->seq_type != 1 && A->seq_type != 2 && A->seq_type != 3)) { rc = -24; break; }
    uint32_t *syms = (uint32_t *)malloc(4 * ((size_t)np * nc + 1));
    int32_t *out = (int32_t *)calloc((size_t)np * nc + 1, sizeof(int32_t));
    A->vals = out;
    A->corner_to_entry = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)nf);
    memcpy(A->corner_to_entry, faces, sizeof(int32_t) * 3 * (size_t)nf);
    A->sym_begin = r->o;
    if (compressed == 1) { if (orc_decode_symbols_nc(b, n, &r->o, (uint32_t)(np * nc), nc, syms, NULL)) rc = -25; }
    else if (compressed == 0) {                        /* raw values: byte width, then little-endian values */
      const int nb = r_u8(r); if (nb < 1 || nb > 4) rc = -25;
      for (int i = 0; i < np * nc && !rc; i++) { uint32_t v = 0; for (int k = 0; k < nb; k++) v |= (uint32_t)r_u8(r) << (8 * k); syms[i] = v; }
    } else rc = -24;
    A->sym_end = r->o;
    if (!rc && A->pred_method == -2) { for (int i = 0; i < np * nc; i++) out[i] = sgn_sym(syms[i]); }
    else if (!rc && A->pred_method == 0 && A->transform == 1) {
      const int32_t lo = r_i32(r), hi = r_i32(r);
      for (int p = 0; p < np; p++) for (int k = 0; k < nc; k++) out[p * nc + k] = wrap_orig(p ? out[(p - 1) * nc + k] : 0, sgn_sym(syms[p * nc + k]), lo, hi);
    } else if (!rc && A->pred_method == 0 && A->transform == 3 && nc == 2) {
      const int32_t maxq = r_i32(r), cen = r_i32(r);
      int q = 0; while ((1 << q) - 1 < maxq) q++;
      octb ot; oct_init(&ot, q);
      if (ot.MAXQ != maxq || ot.CEN != cen) rc = -30;
      for (int p = 0; p < np && !rc; p++) {
        const int32_t zero[2] = {0, 0}; int32_t corr[2] = { (int32_t)syms[2 * p], (int32_t)syms[2 * p + 1] };
        oct_orig_value(&ot, p ? out + 2 * (p - 1) : zero, corr, out + 2 * p);
      }
    } else if (!rc) rc = -31;
    A->sec_end = r->o;
    free(syms);
    if (r->err) rc = -32;
  }
  for (int d = 0; d < natt && !rc; d++) {              /* data needed by the portable transforms, attribute by attribute */
    drc_att *A = &m->att[d];
    if (A->seq_type == 2) { for (int k = 0; k < A->ncomp && k < 4; k++) A->minv[k] = r_f32(r); A->range = r_f32(r); A->qbits = r_u8(r); }
    else if (A->seq_type == 3) A->qbits = r_u8(r);
    if (r->err) rc = -32;
  }
  free(faces);
  m->total = n; m->leftover = n - r->o;
  if (rc) { m->natt = 8; drc_mesh_free(m); }
  return rc;
}

int drc_decode(const uint8_t *b, size_t n, drc_mesh *m) {
  memset(m, 0, sizeof(*m));
  if (n < 11 || memcmp(b, "DRACO", 5)) return -1;
  m->major = b[5]; m->minor = b[6];
  if (b[7] != 1 || b[8] > 1) return -2;                /* TRIANGULAR_MESH; SEQUENTIAL (0) or EDGEBREAKER (1) */
  if (m->major != 2 || m->minor != 2) return -3;
  if ((b[9] | (b[10] << 8)) != 0) return -4;           /* no metadata */
  if (b[8] == 0) return drc_decode_sequential(b, n, m);
  m->method = 1;
  rdr R = { b, n, 11, 0 }, *r = &R;
  int rc = 0;
  int tt = r_u8(r); if (tt != 2 && tt != 0) return -5; /* VALENCE (2) or STANDARD (0) traversal */
  m->traversal = tt;
  int nev = (int)r_varint(r), nf = (int)r_varint(r), nad = r_u8(r);
  int nsym = (int)r_varint(r), nsplit = (int)r_varint(r), nts = (int)r_varint(r);
  if (r->err || nf <= 0 || nad > 7 || nsym > nf || nts > nf) return -6;
  m->nev = nev; m->nf = nf; m->nad = nad; m->nsym = nsym; m->nsplit = nsplit; m->nts = nts;
  int32_t *sp_src = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nts + 1)), *sp_spl = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nts + 1));
  uint8_t *sp_edge = (uint8_t *)malloc((size_t)nts + 1);
  { int last = 0; for (int i = 0; i < nts; i++) { int d = (int)r_varint(r); int src = d + last; int d2 = (int)r_varint(r); sp_src[i] = src; sp_spl[i] = src - d2; last = src; }
    if (r->o + (size_t)(nts + 7) / 8 > n) r->err = 1;
    else { for (int i = 0; i < nts; i++) sp_edge[i] = (b[r->o + (i >> 3)] >> (i & 7)) & 1; if (nts > 0) r->o += (size_t)(nts + 7) / 8; } }
  orc_rabs_dec start_faces, seams[8];
  const uint8_t *sbits = NULL; size_t sbit = 0, sbit_n = 0;      /* standard traversal: the symbols as a size-prefixed LSB-first bit sequence */
  if (tt == 0) { const size_t nb = r_varint(r); if (r->err || r->o + nb > n) { rc = -7; goto fail0; } sbits = b + r->o; sbit_n = 8 * nb; r->o += nb; }
  if (r->err || orc_rabs_open(&start_faces, b, n, r->o)) { rc = -7; goto fail0; }
  r->o = start_faces.end;
  for (int i = 0; i < nad; i++) { if (orc_rabs_open(&seams[i], b, n, r->o)) { rc = -7; goto fail0; } r->o = seams[i].end; }
  uint32_t *ctx[6] = {0}; int cnt[6];
  for (int i = 0; i < 6; i++) cnt[i] = 0;
  for (int i = 0; i < 6 && tt == 2; i++) {
    int cn = (int)r_varint(r); cnt[i] = cn; m->ctx_n[i] = cn;
    if (r->err || cn < 0 || cn > nf) { rc = -8; goto fail1; }
    if (cn > 0) { ctx[i] = (uint32_t *)malloc(4 * (size_t)cn); if (orc_decode_symbols(b, n, &r->o, (uint32_t)cn, ctx[i], NULL)) { rc = -8; goto fail1; } }
  }
  m->conn_end = r->o;
  {
    int maxv = nev + nsplit + 3;
    int32_t *opp = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)nf), *c2v = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)nf);
    int32_t *lm = (int32_t *)malloc(sizeof(int32_t) * (size_t)maxv), *val = (int32_t *)calloc((size_t)maxv, sizeof(int32_t));
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nf + 8));
    int32_t *tsac = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nsym + 1));   /* split map: decoder symbol id -> active corner */
    for (int i = 0; i < 3 * nf; i++) { opp[i] = ORC_INV; c2v[i] = ORC_INV; }
    for (int i = 0; i <= nsym; i++) tsac[i] = ORC_INV;
    m->opp = opp; m->c2v = c2v;
    int nv = 0, sp = 0, nfaces = 0, active_ctx = -1, splits_left = nts;
    static const int SYM2TOPO[5] = { 0, 1, 3, 5, 7 };   /* C,S,L,R,E */
#define SETOPP(a, bb) do { opp[a] = (bb); opp[bb] = (a); } while (0)
#define ADDV() (nv < maxv ? (lm[nv] = ORC_INV, nv++) : (rc = -9, 0))
    for (int sid = 0; sid < nsym && !rc; sid++) {
      int face = nfaces++, check = 0, sym;
      if (tt == 0) {                                    /* 1 bit: C; else two more bits: S 1, L 3, R 5, E 7 */
        if (sbit + 1 > sbit_n) { rc = -10; break; }
        sym = (sbits[sbit >> 3] >> (sbit & 7)) & 1; sbit++;
        if (sym) { if (sbit + 2 > sbit_n) { rc = -10; break; } for (int k = 0; k < 2; k++, sbit++) sym |= ((sbits[sbit >> 3] >> (sbit & 7)) & 1) << (1 + k); }
      }
      else if (active_ctx != -1) { if (--cnt[active_ctx] < 0) { rc = -10; break; } uint32_t s = ctx[active_ctx][cnt[active_ctx]]; if (s > 4) { rc = -10; break; } sym = SYM2TOPO[s]; }
      else sym = 7;
      int corner = 3 * face;
      if (sym == 0) {                                   /* C */
        if (sp == 0) { rc = -11; break; }
        int ca = stack[sp - 1]; int vx = c2v[c_nxt(ca)]; if (vx < 0 || lm[vx] < 0) { rc = -11; break; }
        int cb = c_nxt(lm[vx]);
        if (ca == cb || opp[ca] != ORC_INV || opp[cb] != ORC_INV) { rc = -11; break; }
        SETOPP(ca, corner + 1); SETOPP(cb, corner + 2);
        int vap = c2v[c_prv(ca)], vbn = c2v[c_nxt(cb)];
        c2v[corner] = vx; c2v[corner + 1] = vbn; c2v[corner + 2] = vap; lm[vap] = corner + 2;
        stack[sp - 1] = corner;
      } else if (sym == 5 || sym == 3) {                /* R / L */
        if (sp == 0) { rc = -12; break; }
        int ca = stack[sp - 1]; if (opp[ca] != ORC_INV) { rc = -12; break; }
        int oc, cl, cr;
        if (sym == 5) { oc = corner + 2; cl = corner + 1; cr = corner; } else { oc = corner + 1; cl = corner; cr = corner + 2; }
        SETOPP(oc, ca); int nvx = ADDV(); if (rc) break; c2v[oc] = nvx; lm[nvx] = oc;
        int vr = c2v[c_prv(ca)]; c2v[cr] = vr; lm[vr] = cr;
        c2v[cl] = c2v[c_nxt(ca)];
        stack[sp - 1] = corner; check = 1;
      } else if (sym == 1) {                            /* S */
        if (sp == 0) { rc = -13; break; }
        int cb = stack[--sp];
        if (tsac[sid] != ORC_INV) stack[sp++] = tsac[sid];
        if (sp == 0) { rc = -13; break; }
        int ca = stack[sp - 1];
        if (ca == cb || opp[ca] != ORC_INV || opp[cb] != ORC_INV) { rc = -13; break; }
        SETOPP(ca, corner + 2); SETOPP(cb, corner + 1);
        int vp = c2v[c_prv(ca)]; c2v[corner] = vp; c2v[corner + 1] = c2v[c_nxt(ca)];
        int vbp = c2v[c_prv(cb)]; c2v[corner + 2] = vbp; lm[vbp] = corner + 2;
        int cn = c_nxt(cb), vn = c2v[cn];
        val[vp] += val[vn]; lm[vp] = lm[vn];
        int first = cn, guard = 0;
        while (cn != ORC_INV) { c2v[cn] = vp; int o2 = opp[c_nxt(cn)]; cn = o2 < 0 ? ORC_INV : c_nxt(o2); if (cn == first || ++guard > 3 * nf) { rc = -13; break; } }
        if (rc) break;
        lm[vn] = ORC_INV;
        stack[sp - 1] = corner;
      } else {                                          /* E */
        int v0 = ADDV(), v1 = ADDV(), v2 = ADDV(); if (rc) break;
        c2v[corner] = v0; c2v[corner + 1] = v1; c2v[corner + 2] = v2; lm[v0] = corner; lm[v1] = corner + 1; lm[v2] = corner + 2;
        stack[sp++] = corner; check = 1;
      }
      { int c = stack[sp - 1], nn = c_nxt(c), pp = c_prv(c);
        if (sym == 0 || sym == 1) { val[c2v[nn]] += 1; val[c2v[pp]] += 1; }
        else if (sym == 5) { val[c2v[c]] += 1; val[c2v[nn]] += 1; val[c2v[pp]] += 2; }
        else if (sym == 3) { val[c2v[c]] += 1; val[c2v[nn]] += 2; val[c2v[pp]] += 1; }
        else { val[c2v[c]] += 2; val[c2v[nn]] += 2; val[c2v[pp]] += 2; }
        int av = val[c2v[nn]]; av = av < 2 ? 2 : (av > 7 ? 7 : av); active_ctx = av - 2; }
      if (check) {
        int esid = nsym - sid - 1;
        while (splits_left > 0 && sp_src[splits_left - 1] == esid) {
          splits_left--;
          int top = stack[sp - 1];
          int nac = sp_edge[splits_left] == 1 ? c_nxt(top) : c_prv(top);
          int dsid = nsym - sp_spl[splits_left] - 1;
          if (dsid < 0 || dsid > nsym) { rc = -14; break; }
          tsac[dsid] = nac;
        }
      }
    }
    while (!rc && sp > 0) {
      int corner = stack[--sp];
      if (orc_rabs_bit(&start_faces)) {
        int vn = c2v[c_nxt(corner)]; int cb = c_nxt(lm[vn]); int vx = c2v[c_nxt(cb)]; int cc = c_nxt(lm[vx]);
        int vp = c2v[c_nxt(cc)];
        if (nfaces >= nf) { rc = -15; break; }
        int face = nfaces++, nc = 3 * face;
        SETOPP(nc, corner); SETOPP(nc + 1, cb); SETOPP(nc + 2, cc);
        c2v[nc] = vx; c2v[nc + 1] = vp; c2v[nc + 2] = vn;
        m->n_interior_start++;
      }
    }
    if (!rc && nfaces != nf) rc = -16;
    for (int i = 0; i < 6 && !rc; i++) if (cnt[i] != 0) rc = -17;
    m->nverts_alloc = nv;
    if (rc) { free(lm); free(val); free(stack); free(tsac); goto fail1; }

    /* ---- seams (A.5) ---- */
    uint8_t *edge_seam[8] = {0};
    for (int i = 0; i < nad; i++) edge_seam[i] = (uint8_t *)calloc(3 * (size_t)nf, 1);
    uint32_t nseam[8] = {0};
    for (int f = 0; f < nf; f++) for (int k = 0; k < 3; k++) {
      int c = 3 * f + k, oc = opp[c];
      if (oc == ORC_INV) { for (int i = 0; i < nad; i++) { edge_seam[i][c] = 1; nseam[i]++; } continue; }
      if (oc / 3 < f) continue;
      for (int i = 0; i < nad; i++) if (orc_rabs_bit(&seams[i])) { edge_seam[i][c] = 1; edge_seam[i][oc] = 1; nseam[i]++; }
    }
    /* attribute corner tables */
    int32_t *t_c2v[8] = {0}, *t_lm[8] = {0}; int t_nv[8] = {0};
    for (int i = 0; i < nad; i++) {
      uint8_t *vseam = (uint8_t *)calloc((size_t)nv + 1, 1);
      for (int c = 0; c < 3 * nf; c++) if (edge_seam[i][c]) { vseam[c2v[c_nxt(c)]] = 1; vseam[c2v[c_prv(c)]] = 1; }
      t_c2v[i] = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)nf); t_lm[i] = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)nf);
      for (int c = 0; c < 3 * nf; c++) t_c2v[i][c] = ORC_INV;
      ctab T = { nf, 0, opp, edge_seam[i], t_c2v[i], t_lm[i] };
      int tn = 0;
      for (int v = 0; v < nv; v++) {
        int c = lm[v]; if (c == ORC_INV) continue;
        int vid = tn, first = c;
        if (vseam[v]) { int a = t_swing_left(&T, first); while (a != ORC_INV) { first = a; a = t_swing_left(&T, a); if (a == c) break; } }
        t_c2v[i][first] = vid; t_lm[i][tn++] = first;
        int a = (opp[c_prv(first)] == ORC_INV) ? ORC_INV : c_prv(opp[c_prv(first)]);
        while (a != ORC_INV && a != first) {
          if (edge_seam[i][c_nxt(a)]) { vid = tn; t_lm[i][tn++] = a; }
          t_c2v[i][a] = vid;
          a = (opp[c_prv(a)] == ORC_INV) ? ORC_INV : c_prv(opp[c_prv(a)]);
        }
      }
      t_nv[i] = tn; free(vseam);
    }

    /* ---- attribute decoder headers (A.4) ---- */
    int ndec = r_u8(r); if (r->err || ndec < 1 || ndec > 8) { rc = -20; goto fail2; }
    for (int d = 0; d < ndec; d++) { m->att[d].att_data_id = (int8_t)r_u8(r); m->att[d].dec_type = r_u8(r); int trav = r_u8(r); if (trav != 0) { rc = -21; goto fail2; } }
    for (int d = 0; d < ndec; d++) {
      int na = (int)r_varint(r); if (na != 1) { rc = -22; goto fail2; }
      m->att[d].att_type = r_u8(r); m->att[d].data_type = r_u8(r); m->att[d].ncomp = r_u8(r); (void)r_u8(r); m->att[d].unique_id = (int)r_varint(r);
      m->att[d].seq_type = r_u8(r);
    }
    m->natt = ndec; m->hdr_end = r->o;
    if (r->err) { rc = -22; goto fail2; }

    /* base traversal (shared by every vertex-type attribute) */
    int32_t *b_order = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)nf), *b_v2d = (int32_t *)malloc(sizeof(int32_t) * ((size_t)nv + 1));
    ctab B = { nf, nv, opp, NULL, c2v, lm };
    int b_n = traverse(&B, b_order, b_v2d);
    const int32_t *P = NULL;    /* quantised positions, entry order */

    for (int d = 0; d < ndec && !rc; d++) {
      drc_att *A = &m->att[d];
      ctab X = B; const int32_t *order = b_order, *v2d = b_v2d; int ne = b_n;
      int32_t *own_order = NULL, *own_v2d = NULL;
      if (A->dec_type == 1) {
        int ad = A->att_data_id; if (ad < 0 || ad >= nad) { rc = -23; break; }
        X.edge_seam = edge_seam[ad]; X.c2v = t_c2v[ad]; X.lm = t_lm[ad]; X.nverts = t_nv[ad];
        own_order = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)nf); own_v2d = (int32_t *)malloc(sizeof(int32_t) * ((size_t)t_nv[ad] + 1));
        ne = traverse(&X, own_order, own_v2d); order = own_order; v2d = own_v2d;
        A->n_seam_corners = nseam[ad];
      }
      A->n = ne; A->sec_begin = r->o;
      A->pred_method = (int8_t)r_u8(r); A->transform = (int8_t)r_u8(r); int compressed = r_u8(r);
      if (compressed != 1) { rc = -24; }
      int nc = (A->seq_type == 3) ? 2 : A->ncomp; A->ncomp_port = nc;
      uint32_t *syms = (uint32_t *)malloc(4 * ((size_t)ne * nc + 1));
      orc_sym_info si; A->sym_begin = r->o;
      if (!rc && orc_decode_symbols(b, n, &r->o, (uint32_t)(ne * nc), syms, &si)) rc = -25;
      A->sym_end = r->o;
      int32_t *out = (int32_t *)calloc((size_t)ne * nc + 1, sizeof(int32_t));
      A->vals = out;
      A->corner_to_entry = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)nf);
      for (int c = 0; c < 3 * nf; c++) A->corner_to_entry[c] = v2d[X.c2v[c]];
      if (!rc && (A->pred_method == 1 || A->pred_method == 0) && A->transform == 1) {
        int32_t lo = r_i32(r), hi = r_i32(r);
        for (int p = 0; p < ne; p++) {
          int32_t pred[4] = {0, 0, 0, 0}; int have = 0;
          if (p > 0 && A->pred_method == 1) {
            int ci = order[p], oci = t_opp(&X, ci);
            if (oci != ORC_INV) {
              int a = v2d[X.c2v[oci]], bn = v2d[X.c2v[c_nxt(oci)]], bp = v2d[X.c2v[c_prv(oci)]];
              if (a < p && bn < p && bp < p) { for (int k = 0; k < nc; k++) pred[k] = out[bn * nc + k] + out[bp * nc + k] - out[a * nc + k]; have = 1; }
            }
          }
          if (!have && p > 0) for (int k = 0; k < nc; k++) pred[k] = out[(p - 1) * nc + k];
          for (int k = 0; k < nc; k++) out[p * nc + k] = wrap_orig(pred[k], sgn_sym(syms[p * nc + k]), lo, hi);
        }
      } else if (!rc && A->pred_method == 5 && A->transform == 1 && nc == 2) {
        if (!P) { rc = -26; }
        int32_t no = r_i32(r);
        orc_rabs_dec Rb; if (!rc && (no < 0 || orc_rabs_open(&Rb, b, n, r->o))) rc = -26;
        uint8_t *ori = (uint8_t *)malloc((size_t)(no > 0 ? no : 1)); int last = 1;
        if (!rc) { for (int k = 0; k < no; k++) { if (!orc_rabs_bit(&Rb)) last = !last; ori[k] = (uint8_t)last; } r->o = Rb.end; }
        A->n_orient = no;
        int32_t lo = r_i32(r), hi = r_i32(r); int nori = no;
        for (int p = 0; p < ne && !rc; p++) {
          int c = order[p], cn = c_nxt(c), cp = c_prv(c);
          int nd = v2d[X.c2v[cn]], pd = v2d[X.c2v[cp]];
          int64_t pred[2]; int have = 0;
          if (pd < p && nd < p) {
            int64_t nuv[2] = { out[nd * 2], out[nd * 2 + 1] }, puv[2] = { out[pd * 2], out[pd * 2 + 1] };
            if (puv[0] == nuv[0] && puv[1] == nuv[1]) { pred[0] = puv[0]; pred[1] = puv[1]; have = 1; }
            else {
              const int32_t *tip = P + 3 * b_v2d[c2v[c]], *np_ = P + 3 * b_v2d[c2v[cn]], *pp_ = P + 3 * b_v2d[c2v[cp]];
              int64_t pn[3], cnv[3], pn2 = 0, dd = 0;
              for (int k = 0; k < 3; k++) { pn[k] = (int64_t)pp_[k] - np_[k]; pn2 += pn[k] * pn[k]; }
              if (pn2 != 0) {
                for (int k = 0; k < 3; k++) { cnv[k] = (int64_t)tip[k] - np_[k]; dd += pn[k] * cnv[k]; }
                int64_t pnuv[2] = { puv[0] - nuv[0], puv[1] - nuv[1] };
                int64_t xuv[2] = { nuv[0] * pn2 + dd * pnuv[0], nuv[1] * pn2 + dd * pnuv[1] };
                int64_t cx2 = 0;
                for (int k = 0; k < 3; k++) { int64_t xp = np_[k] + (dd * pn[k]) / pn2; int64_t e = tip[k] - xp; cx2 += e * e; }
                int64_t ns_ = (int64_t)orc_isqrt((uint64_t)cx2 * (uint64_t)pn2);
                int64_t cxuv[2] = { pnuv[1] * ns_, -pnuv[0] * ns_ };
                if (nori <= 0) { rc = -27; break; }
                int o_ = ori[--nori];
                if (o_) { pred[0] = (xuv[0] + cxuv[0]) / pn2; pred[1] = (xuv[1] + cxuv[1]) / pn2; }
                else { pred[0] = (xuv[0] - cxuv[0]) / pn2; pred[1] = (xuv[1] - cxuv[1]) / pn2; }
                have = 1;
              }
            }
          }
          if (!have) {
            if (nd < p) { pred[0] = out[nd * 2]; pred[1] = out[nd * 2 + 1]; }
            else if (p > 0) { pred[0] = out[(p - 1) * 2]; pred[1] = out[(p - 1) * 2 + 1]; }
            else { pred[0] = pred[1] = 0; }
          }
          for (int k = 0; k < 2; k++) out[p * 2 + k] = wrap_orig((int32_t)pred[k], sgn_sym(syms[p * 2 + k]), lo, hi);
        }
        if (!rc && nori != 0) rc = -28;
        free(ori);
      } else if (!rc && A->pred_method == 6 && A->transform == 3 && nc == 2) {
        if (!P) { rc = -29; }
        int32_t maxq = r_i32(r), cen = r_i32(r);
        orc_rabs_dec Fb; if (!rc && orc_rabs_open(&Fb, b, n, r->o)) rc = -29;
        if (!rc) r->o = Fb.end;
        int q = 0; while ((1 << q) - 1 < maxq) q++;
        octb ot; oct_init(&ot, q);
        if (!rc && (ot.MAXQ != maxq || ot.CEN != cen)) rc = -30;
        for (int dd = 0; dd < ne && !rc; dd++) {
          int c0 = order[dd];
          const int32_t *cenp = P + 3 * b_v2d[c2v[c0]];
          int64_t N[3] = {0, 0, 0};
          /* VertexCornersIterator over the attribute-vertex fan */
          int c = c0, left = 1;
          while (c != ORC_INV) {
            const int32_t *a = P + 3 * b_v2d[c2v[c_nxt(c)]], *bb = P + 3 * b_v2d[c2v[c_prv(c)]];
            int64_t dn[3], dp[3];
            for (int k = 0; k < 3; k++) { dn[k] = (int64_t)a[k] - cenp[k]; dp[k] = (int64_t)bb[k] - cenp[k]; }
            N[0] += dn[1] * dp[2] - dn[2] * dp[1]; N[1] += dn[2] * dp[0] - dn[0] * dp[2]; N[2] += dn[0] * dp[1] - dn[1] * dp[0];
            if (left) { c = t_swing_left(&X, c); if (c == c0) break; if (c == ORC_INV) { left = 0; c = t_swing_right(&X, c0); } }
            else c = t_swing_right(&X, c);
          }
          int64_t s = llabs(N[0]) + llabs(N[1]) + llabs(N[2]);
          if (s > (1 << 29)) { int64_t qd = s / (1 << 29); for (int k = 0; k < 3; k++) N[k] /= qd; }
          int32_t pv[3]; oct_canon_vec(&ot, N, pv);
          if (orc_rabs_bit(&Fb)) { pv[0] = -pv[0]; pv[1] = -pv[1]; pv[2] = -pv[2]; A->n_flip_set++; }
          int32_t po[2]; oct_vec_to_oct(&ot, pv, &po[0], &po[1]);
          int32_t corr[2] = { (int32_t)syms[2 * dd], (int32_t)syms[2 * dd + 1] };
          oct_orig_value(&ot, po, corr, out + 2 * dd);
        }
      } else if (!rc) rc = -31;
      A->sec_end = r->o;
      /* data needed by portable transforms */
      if (!rc) {
        if (A->seq_type == 2) { for (int k = 0; k < A->ncomp; k++) A->minv[k] = r_f32(r); A->range = r_f32(r); A->qbits = r_u8(r); }
        else if (A->seq_type == 3) { A->qbits = r_u8(r); }
      }
      if (r->err) rc = -32;
      if (!rc && A->att_type == 0 && A->att_data_id == -1) P = out;
      free(syms); free(own_order); free(own_v2d);
    }
    m->total = n; m->leftover = n - r->o;
    free(b_order); free(b_v2d);
fail2:
    for (int i = 0; i < nad; i++) { free(edge_seam[i]); free(t_c2v[i]); free(t_lm[i]); }
    free(lm); free(val); free(stack); free(tsac);
  }
fail1:
  for (int i = 0; i < 6; i++) free(ctx[i]);
fail0:
  free(sp_src); free(sp_spl); free(sp_edge);
  if (rc) { int natt = m->natt; m->natt = 8; (void)natt; drc_mesh_free(m); }
  return rc;
}

void drc_dequant(const drc_mesh *m, int a, float *out) {
  const drc_att *A = &m->att[a];
  if (A->seq_type == 2) {
    float delta = A->range / (float)((1u << A->qbits) - 1);
    for (int i = 0; i < A->n; i++) for (int k = 0; k < A->ncomp; k++) out[i * A->ncomp + k] = A->minv[k] + (float)A->vals[i * A->ncomp + k] * delta;
  } else if (A->seq_type == 3) {
    octb t; oct_init(&t, A->qbits);
    for (int i = 0; i < A->n; i++) {
      float y = (float)A->vals[2 * i] * (2.0f / (float)t.MAXV) - 1.0f, z = (float)A->vals[2 * i + 1] * (2.0f / (float)t.MAXV) - 1.0f;
      float x = 1.0f - fabsf(y) - fabsf(z), xo = x < 0 ? -x : 0;
      y += y < 0 ? xo : -xo; z += z < 0 ? xo : -xo;
      float nn = sqrtf(x * x + y * y + z * z);
      if (nn > 1e-6f) { x /= nn; y /= nn; z /= nn; } else { x = y = z = 0; }
      out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z;
    }
  } else {
    for (int i = 0; i < A->n * A->ncomp; i++) out[i] = (float)A->vals[i];
  }
}
