/* oracle/ktx2_enc.c — TEST INFRASTRUCTURE (see oracle_common.h).
 * CPU restatement of the texture half of the hot path: what one
 *   basisu -ktx2 -tex_type video -multifile_num B -y_flip        (scripts/Encoder.py:290)
 * process computes — ETC1S endpoint/selector search, global codebooks, P-frame skip blocks,
 * BasisLZ Huffman slices, KTX2 container (SURVEY.md B.0–B.4; encoder side B.6 is "any encoder that
 * emits valid B.0–B.3 structures is playable").  basis_universal's own clustering is not vendored in
 * /root/reference and cannot be restated byte-for-byte; this file defines a deterministic,
 * integer-only algorithm (documented step by step in DESIGN.md §texture) that the HIP path
 * reproduces bit-exactly, and whose output is validated by decoding with ktx2_dec.c (pinned on the
 * 50 reference fixtures) and by PSNR/bpp against the fixtures' operating point.
 */
#include "ktx2_oracle.h"
#include <stdio.h>

static const int INTEN[8][4] = { {-8, -2, 2, 8}, {-17, -5, 5, 17}, {-29, -9, 9, 29}, {-42, -13, 13, 42}, {-60, -18, 18, 60}, {-80, -24, 24, 80}, {-106, -33, 33, 106}, {-183, -47, 47, 183} };
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int expand5(int c) { return (c << 3) | (c >> 2); }

/* ------------------------------------------------------------------ bit writer (LSB first) */
typedef struct { orc_buf b; uint64_t acc; int nacc; } bitw;
static void bw_put(bitw *w, uint32_t v, int n) {
  while (n > 0) { int k = n > 24 ? 24 : n; w->acc |= (uint64_t)(v & ((1u << k) - 1)) << w->nacc; w->nacc += k; v >>= k; n -= k;
    while (w->nacc >= 8) { ob_u8(&w->b, (uint8_t)(w->acc & 0xff)); w->acc >>= 8; w->nacc -= 8; } }
}
static void bw_flush(bitw *w) { if (w->nacc > 0) { ob_u8(&w->b, (uint8_t)(w->acc & 0xff)); w->acc = 0; w->nacc = 0; } }
static void bw_vlc(bitw *w, uint32_t v, int cb) {
  for (;;) { uint32_t chunk = v & ((1u << cb) - 1); v >>= cb; bw_put(w, chunk | (v ? (1u << cb) : 0), cb + 1); if (!v) break; }
}

/* ------------------------------------------------------------------ Huffman (length-limited, canonical) */
typedef struct { int n; uint8_t *size; uint16_t *code; } hcode;     /* code is stored bit-reversed, ready for LSB-first emission */
static void hcode_free(hcode *h) { free(h->size); free(h->code); h->size = NULL; h->code = NULL; }
typedef struct { uint32_t f; uint32_t s; } hsym;
static int cmp_hsym(const void *a, const void *b) { const hsym *x = (const hsym *)a, *y = (const hsym *)b; if (x->f != y->f) return x->f < y->f ? -1 : 1; return x->s < y->s ? -1 : (x->s > y->s); }
/* freq[n] -> code lengths (<= maxlen) + canonical codes.  Deterministic: sort (freq asc, sym asc);
 * two-queue merge preferring the leaf on ties; miniz-style Kraft repair for the length limit. */
static void hcode_build(hcode *h, const uint32_t *freq_in, int n, int maxlen) {
  h->n = n; h->size = (uint8_t *)calloc((size_t)n + 1, 1); h->code = (uint16_t *)calloc((size_t)n + 1, 2);
  hsym *sy = (hsym *)malloc(sizeof(hsym) * (size_t)(n + 1)); int m = 0;
  for (int i = 0; i < n; i++) if (freq_in[i]) { sy[m].f = freq_in[i]; sy[m].s = (uint32_t)i; m++; }
  if (m == 0) { sy[0].f = 1; sy[0].s = 0; m = 1; }
  if (m == 1) { h->size[sy[0].s] = 1; }
  else {
    qsort(sy, (size_t)m, sizeof(hsym), cmp_hsym);
    /* two-queue Huffman: nodes 0..m-1 leaves (sorted), m.. internal */
    uint64_t *w = (uint64_t *)malloc(8 * (size_t)(2 * m)); int *parent = (int *)malloc(sizeof(int) * (size_t)(2 * m));
    for (int i = 0; i < m; i++) w[i] = sy[i].f;
    int li = 0, ni = m, nn = m;
    for (int k = 0; k < m - 1; k++) {
      int a, b;
      if (li < m && (ni >= nn || w[li] <= w[ni])) a = li++; else a = ni++;
      if (li < m && (ni >= nn || w[li] <= w[ni])) b = li++; else b = ni++;
      w[nn] = w[a] + w[b]; parent[a] = nn; parent[b] = nn; nn++;
    }
    parent[nn - 1] = -1;
    int cnt[64]; memset(cnt, 0, sizeof(cnt));
    for (int i = 0; i < m; i++) { int d = 0, p = i; while (parent[p] >= 0) { p = parent[p]; d++; } if (d > 63) d = 63; cnt[d]++; }
    /* enforce max code size */
    for (int l = maxlen + 1; l < 64; l++) { cnt[maxlen] += cnt[l]; cnt[l] = 0; }
    uint64_t total = 0; for (int l = maxlen; l > 0; l--) total += (uint64_t)cnt[l] << (maxlen - l);
    while (total != (1ull << maxlen)) {
      cnt[maxlen]--;
      for (int l = maxlen - 1; l > 0; l--) if (cnt[l]) { cnt[l]--; cnt[l + 1] += 2; break; }
      total--;
    }
    /* least frequent symbols get the longest codes */
    int j = 0; for (int l = maxlen; l >= 1; l--) for (int c = 0; c < cnt[l]; c++) h->size[sy[j++].s] = (uint8_t)l;
    free(w); free(parent);
  }
  /* canonical codes */
  uint32_t blc[20]; memset(blc, 0, sizeof(blc));
  for (int i = 0; i < n; i++) if (h->size[i]) blc[h->size[i]]++;
  uint32_t next[20]; uint32_t code = 0; next[0] = 0;
  for (int l = 1; l <= 16; l++) { code = (code + blc[l - 1]) << 1; next[l] = code; }
  for (int i = 0; i < n; i++) if (h->size[i]) {
    uint32_t c = next[h->size[i]]++, r = 0; for (int k = 0; k < h->size[i]; k++) r |= ((c >> k) & 1) << (h->size[i] - 1 - k);
    h->code[i] = (uint16_t)r;
  }
  free(sy);
}
static inline void hcode_put(bitw *w, const hcode *h, uint32_t s) { bw_put(w, h->code[s], h->size[s]); }

static const int ZZ[21] = { 17, 18, 19, 20, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15, 16 };
/* serialise a code-length table (SURVEY B.1 read_huff, inverse) */
static void write_huff(bitw *w, const hcode *h) {
  int total = 0; for (int i = 0; i < h->n; i++) if (h->size[i]) total = i + 1;
  bw_put(w, (uint32_t)total, 14);
  if (!total) return;
  /* RLE tokens */
  uint8_t *tok = (uint8_t *)malloc((size_t)total + 8); uint8_t *ext = (uint8_t *)malloc((size_t)total + 8); int nt = 0;
  for (int i = 0; i < total;) {
    int len = h->size[i], run = 1; while (i + run < total && h->size[i + run] == len) run++;
    i += run;
    if (len == 0) {
      while (run > 0) {
        if (run < 3) { tok[nt] = 0; ext[nt++] = 0; run--; }
        else if (run <= 10) { tok[nt] = 17; ext[nt++] = (uint8_t)(run - 3); run = 0; }
        else { int r = run > 138 ? 138 : run; tok[nt] = 18; ext[nt++] = (uint8_t)(r - 11); run -= r; }
      }
    } else {
      tok[nt] = (uint8_t)len; ext[nt++] = 0; run--;
      while (run > 0) {
        if (run < 3) { tok[nt] = (uint8_t)len; ext[nt++] = 0; run--; }
        else if (run <= 6) { tok[nt] = 19; ext[nt++] = (uint8_t)(run - 3); run = 0; }
        else { int r = run > 134 ? 134 : run; tok[nt] = 20; ext[nt++] = (uint8_t)(r - 7); run -= r; }
      }
    }
  }
  uint32_t f[21]; memset(f, 0, sizeof(f)); for (int i = 0; i < nt; i++) f[tok[i]]++;
  hcode cl; hcode_build(&cl, f, 21, 7);
  int ncl = 1; for (int i = 0; i < 21; i++) if (cl.size[ZZ[i]]) ncl = i + 1;
  bw_put(w, (uint32_t)ncl, 5);
  for (int i = 0; i < ncl; i++) bw_put(w, cl.size[ZZ[i]], 3);
  for (int i = 0; i < nt; i++) {
    hcode_put(w, &cl, tok[i]);
    if (tok[i] == 17) bw_put(w, ext[i], 3); else if (tok[i] == 18) bw_put(w, ext[i], 7); else if (tok[i] == 19) bw_put(w, ext[i], 2); else if (tok[i] == 20) bw_put(w, ext[i], 7);
  }
  hcode_free(&cl); free(tok); free(ext);
}

/* ------------------------------------------------------------------ tree-structured VQ (level synchronous) */
#define TSVQ_MAX_ROUNDS 24
typedef struct { int64_t W, S[16], Q[16]; } lstat;
/* items: n points of `dim` small non-negative ints (x[i*dim+d]) with weight w[i]; dim weights wd[].
 * On return leaf[i] in [0, *nleaves). */
static void tsvq(const int32_t *x, const uint32_t *w, uint32_t n, int dim, const int *wd, uint32_t K, uint32_t *leaf, uint32_t *nleaves_out) {
  uint32_t nl = 1;
  for (uint32_t i = 0; i < n; i++) leaf[i] = 0;
  lstat *st = (lstat *)malloc(sizeof(lstat) * (size_t)(K + 1));
  uint8_t *split = (uint8_t *)malloc(K + 1), *chosen = (uint8_t *)malloc(K + 1); int *axis = (int *)malloc(sizeof(int) * (K + 1)); int64_t *th = (int64_t *)malloc(8 * (size_t)(K + 1)), *prio = (int64_t *)malloc(8 * (size_t)(K + 1));
  uint32_t *newidx = (uint32_t *)malloc(4 * (size_t)(K + 1));
  for (int round = 0; round < TSVQ_MAX_ROUNDS && nl < K; round++) {
    memset(st, 0, sizeof(lstat) * nl);
    for (uint32_t i = 0; i < n; i++) { lstat *s = &st[leaf[i]]; int64_t wi = w ? w[i] : 1; s->W += wi; for (int d = 0; d < dim; d++) { int64_t v = x[(size_t)i * dim + d]; s->S[d] += wi * v; s->Q[d] += wi * v * v; } }
    uint32_t navail = 0;
    for (uint32_t l = 0; l < nl; l++) {
      const lstat *s = &st[l]; int64_t D = 0, best = -1; int ax = 0;
      for (int d = 0; d < dim; d++) { int64_t num = (s->W * s->Q[d] - s->S[d] * s->S[d]) * wd[d]; D += num; if (num > best) { best = num; ax = d; } }
      split[l] = D > 0; axis[l] = ax; th[l] = s->W ? s->S[ax] / s->W : 0; prio[l] = s->W ? D / s->W : 0;
      navail += split[l];
    }
    if (!navail) break;
    uint32_t m = navail < K - nl ? navail : K - nl;
    if (navail <= K - nl) memcpy(chosen, split, nl);
    else for (uint32_t l = 0; l < nl; l++) {
      chosen[l] = 0; if (!split[l]) continue;
      uint32_t rank = 0; for (uint32_t j = 0; j < nl; j++) if (split[j] && (prio[j] > prio[l] || (prio[j] == prio[l] && j < l))) rank++;
      chosen[l] = rank < m;
    }
    uint32_t c = 0; for (uint32_t l = 0; l < nl; l++) { newidx[l] = nl + c; c += chosen[l]; }
    for (uint32_t i = 0; i < n; i++) { uint32_t l = leaf[i]; if (chosen[l] && x[(size_t)i * dim + axis[l]] > th[l]) leaf[i] = newidx[l]; }
    nl += m;
  }
  *nleaves_out = nl;
  free(st); free(split); free(chosen); free(axis); free(th); free(prio); free(newidx);
}

/* ------------------------------------------------------------------ block model */
typedef struct { uint8_t px[16][3]; } blk;
static uint32_t eval_block(const blk *b, const int c5[3], int t, uint32_t *sel_out, uint16_t etab[16][4]) {
  int base[3] = { expand5(c5[0]), expand5(c5[1]), expand5(c5[2]) };
  int col[4][3]; for (int s = 0; s < 4; s++) for (int c = 0; c < 3; c++) col[s][c] = clampi(base[c] + INTEN[t][s], 0, 255);
  uint32_t tot = 0, sel = 0;
  for (int i = 0; i < 16; i++) {
    uint32_t be = 0xffffffffu; int bs = 0;
    for (int s = 0; s < 4; s++) { uint32_t e = 0; for (int c = 0; c < 3; c++) { int d = col[s][c] - b->px[i][c]; e += (uint32_t)(d * d); } if (etab) etab[i][s] = (uint16_t)(e > 65535 ? 65535 : e); if (e < be) { be = e; bs = s; } }
    tot += be; sel |= (uint32_t)bs << (2 * i);     /* texel i = y*4+x -> byte y, bits 2x */
  }
  if (sel_out) *sel_out = sel;
  return tot;
}

static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : (x > y); }
static uint32_t uniq_sorted(uint32_t *a, uint32_t n) { if (!n) return 0; uint32_t m = 1; for (uint32_t i = 1; i < n; i++) if (a[i] != a[m - 1]) a[m++] = a[i]; return m; }
static uint32_t bsearch_u32(const uint32_t *a, uint32_t n, uint32_t key) { uint32_t lo = 0, hi = n; while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (a[mid] < key) lo = mid + 1; else hi = mid; } return lo; }

static void cell_coords(uint32_t cell, int32_t x[4]) { x[0] = expand5((cell >> 10) & 31); x[1] = expand5((cell >> 5) & 31); x[2] = expand5(cell & 31); x[3] = INTEN[(cell >> 15) & 7][3]; }

int ktx2_encode(const uint8_t *const *layers, int Lr, uint32_t W, uint32_t H, const ktx2_enc_params *prm, orc_buf *out) {
  if (Lr < 1 || Lr > KTX2_MAX_LAYERS || W == 0 || H == 0 || W > 16384 || H > 16384) return -1;
  /* Alpha (basisu: any source image with alpha != 255 gives the file alpha slices; KTX2Loader.js:493-497 reads them): every
   * image then contributes TWO slices, colour and alpha, in the order rgb0 a0 rgb1 a1 ...; the alpha slice is the image
   * (a, a, a) run through the same block model, the same two codebooks and the same Huffman tables; a P-frame slice refers
   * to the previous slice OF ITS KIND.  Below, `L` counts slices ("virtual layers"): slice v shows image v / stride,
   * kind v % stride (0 colour, 1 alpha), and its predecessor is slice v - stride.  PARITY UNPINNED: no reference fixture has alpha. */
  int has_alpha = 0;
  for (int l = 0; l < Lr && !has_alpha; l++) for (size_t i = 0; i < (size_t)W * H; i++) if (layers[l][4 * i + 3] != 255) { has_alpha = 1; break; }
  const int stride = has_alpha ? 2 : 1, L = Lr * stride;
  if (L > KTX2_MAX_LAYERS) return -1;
  const int q = clampi(prm && prm->quality > 0 ? prm->quality : 128, 1, 255), yflip = prm ? prm->y_flip : 1;
  const uint32_t bx = (W + 3) / 4, by = (H + 3) / 4, nb = bx * by, NB = nb * (uint32_t)L;
  const uint32_t Kmax_e = (uint32_t)clampi(q * 12, 32, 16128), Kmax_s = (uint32_t)clampi(q * 6, 32, 16128);
  const uint32_t T_skip = (uint32_t)((255 - q) * 3 / 2);
  /* ---- fetch blocks (y flip + edge replicate) ---- */
  blk *B = (blk *)malloc(sizeof(blk) * (size_t)NB);
  for (int l = 0; l < L; l++) for (uint32_t Y = 0; Y < by; Y++) for (uint32_t X = 0; X < bx; X++) {
    blk *b = &B[(size_t)l * nb + Y * bx + X];
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
      uint32_t px = X * 4 + x, py = Y * 4 + y; if (px >= W) px = W - 1; if (py >= H) py = H - 1;
      uint32_t sr = yflip ? H - 1 - py : py;
      const uint8_t *p = layers[l / stride] + 4 * ((size_t)sr * W + px);
      if (l % stride) { b->px[y * 4 + x][0] = b->px[y * 4 + x][1] = b->px[y * 4 + x][2] = p[3]; }
      else { b->px[y * 4 + x][0] = p[0]; b->px[y * 4 + x][1] = p[1]; b->px[y * 4 + x][2] = p[2]; }
    }
  }
  /* ---- step 0: P-frame skip flags against the anchor (last coded) source block ---- */
  uint8_t *skip = (uint8_t *)calloc(NB, 1);
  for (uint32_t b = 0; b < nb; b++) for (int kind = 0; kind < stride; kind++) {
    int anchor = kind;
    for (int l = kind + stride; l < L; l += stride) {
      const blk *c = &B[(size_t)l * nb + b], *a = &B[(size_t)anchor * nb + b]; uint32_t d = 0;
      for (int i = 0; i < 16; i++) for (int k = 0; k < 3; k++) { int e = (int)c->px[i][k] - a->px[i][k]; d += (uint32_t)(e * e); }
      if (d <= T_skip) skip[(size_t)l * nb + b] = 1; else anchor = l;
    }
  }
  /* ---- step A: per-block endpoint fit ---- */
  uint32_t *cell = (uint32_t *)malloc(4 * (size_t)NB);
  uint32_t *hist = (uint32_t *)calloc(1u << 18, 4);
  for (uint32_t b = 0; b < NB; b++) {
    if (skip[b]) { cell[b] = 0; continue; }
    const blk *bk = &B[b]; int sum[3] = {0, 0, 0};
    for (int i = 0; i < 16; i++) for (int c = 0; c < 3; c++) sum[c] += bk->px[i][c];
    int base5[3]; for (int c = 0; c < 3; c++) base5[c] = (((sum[c] + 8) >> 4) * 31 + 127) / 255;
    uint32_t best = 0xffffffffu; int bc[3] = {0, 0, 0}, bt = 0;
    static const int DS[3] = { 0, -1, 1 };
    for (int t = 0; t < 8; t++) for (int di = 0; di < 3; di++) {
      int c5[3]; for (int c = 0; c < 3; c++) c5[c] = clampi(base5[c] + DS[di], 0, 31);
      uint32_t e = eval_block(bk, c5, t, NULL, NULL);
      if (e < best) { best = e; bt = t; bc[0] = c5[0]; bc[1] = c5[1]; bc[2] = c5[2]; }
    }
    for (int c = 0; c < 3; c++) for (int di = 1; di < 3; di++) {
      int c5[3] = { bc[0], bc[1], bc[2] }; c5[c] += DS[di]; if (c5[c] < 0 || c5[c] > 31) continue;
      uint32_t e = eval_block(bk, c5, bt, NULL, NULL);
      if (e < best) { best = e; bc[0] = c5[0]; bc[1] = c5[1]; bc[2] = c5[2]; }
    }
    cell[b] = ((uint32_t)bt << 15) | ((uint32_t)bc[0] << 10) | ((uint32_t)bc[1] << 5) | (uint32_t)bc[2];
    hist[cell[b]]++;
  }
  /* ---- step B: endpoint codebook (TSVQ + 2 Lloyd iterations in endpoint space) ---- */
  uint32_t ncell = 0; for (uint32_t c = 0; c < (1u << 18); c++) ncell += hist[c] != 0;
  uint32_t *cid = (uint32_t *)malloc(4 * (size_t)(ncell + 1)), *cw = (uint32_t *)malloc(4 * (size_t)(ncell + 1)); int32_t *cx = (int32_t *)malloc(16 * (size_t)(ncell + 1));
  uint32_t *cidx = (uint32_t *)malloc(4u << 18);
  { uint32_t k = 0; for (uint32_t c = 0; c < (1u << 18); c++) if (hist[c]) { cid[k] = c; cw[k] = hist[c]; cell_coords(c, cx + 4 * k); cidx[c] = k; k++; } }
  static const int WD4[4] = { 1, 1, 1, 2 }; static const int WD16[16] = { 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1 };
  const uint32_t Ke_t = ncell < Kmax_e ? ncell : Kmax_e;
  uint32_t *cleaf = (uint32_t *)malloc(4 * (size_t)(ncell + 1)), Ke = 0;
  tsvq(cx, cw, ncell, 4, WD4, Ke_t, cleaf, &Ke);
  uint32_t *ent = (uint32_t *)calloc(Ke + 1, 4);      /* entry tuple as cell id */
  lstat *est = (lstat *)malloc(sizeof(lstat) * (size_t)(Ke + 1));
  for (int it = 0; it <= 2; it++) {
    memset(est, 0, sizeof(lstat) * Ke);
    for (uint32_t i = 0; i < ncell; i++) { lstat *s = &est[cleaf[i]]; s->W += cw[i]; for (int d = 0; d < 4; d++) s->S[d] += (int64_t)cw[i] * cx[4 * i + d]; }
    for (uint32_t k = 0; k < Ke; k++) {
      const lstat *s = &est[k]; if (!s->W) continue;
      int c5[3]; for (int d = 0; d < 3; d++) { int m8 = (int)((2 * s->S[d] + s->W) / (2 * s->W)); c5[d] = clampi((m8 * 31 + 127) / 255, 0, 31); }
      int mi = (int)((2 * s->S[3] + s->W) / (2 * s->W)), bt = 0, bd = 1 << 30;
      for (int t = 0; t < 8; t++) { int d = abs(INTEN[t][3] - mi); if (d < bd) { bd = d; bt = t; } }
      ent[k] = ((uint32_t)bt << 15) | ((uint32_t)c5[0] << 10) | ((uint32_t)c5[1] << 5) | (uint32_t)c5[2];
    }
    if (it == 2) break;
    for (uint32_t i = 0; i < ncell; i++) {
      int64_t bd = INT64_MAX; uint32_t bk = 0;
      for (uint32_t k = 0; k < Ke; k++) { int32_t e[4]; cell_coords(ent[k], e); int64_t d = 0; for (int dd = 0; dd < 4; dd++) { int64_t v = cx[4 * i + dd] - e[dd]; d += v * v * WD4[dd]; } if (d < bd) { bd = d; bk = k; } }
      cleaf[i] = bk;
    }
  }
  /* unique used entries, ascending by tuple */
  uint32_t *ecb = (uint32_t *)malloc(4 * (size_t)(Ke + 1)); uint32_t ne = 0;
  for (uint32_t k = 0; k < Ke; k++) if (est[k].W) ecb[ne++] = ent[k];
  qsort(ecb, ne, 4, cmp_u32); ne = uniq_sorted(ecb, ne);
  /* ---- step C/E: block endpoint index + optimal selectors + error tables ---- */
  uint16_t *bei = (uint16_t *)calloc(NB, 2), *bsi = (uint16_t *)calloc(NB, 2);
  uint32_t *bsel = (uint32_t *)calloc(NB, 4);
  uint16_t (*etab)[16][4] = (uint16_t (*)[16][4])malloc(sizeof(uint16_t[16][4]) * (size_t)NB);
  uint32_t nitems = 0; uint32_t *item = (uint32_t *)malloc(4 * (size_t)NB);
  for (uint32_t b = 0; b < NB; b++) {
    if (skip[b]) continue;
    uint32_t tup = ent[cleaf[cidx[cell[b]]]];
    bei[b] = (uint16_t)bsearch_u32(ecb, ne, tup);
    int c5[3] = { (int)((tup >> 10) & 31), (int)((tup >> 5) & 31), (int)(tup & 31) };
    eval_block(&B[b], c5, (int)(tup >> 15), &bsel[b], etab[b]);
    item[nitems++] = b;
  }
  /* ---- step G: selector codebook (TSVQ in 16-D, assignment by true SSE through the error tables) ---- */
  int32_t *sx = (int32_t *)malloc(64 * (size_t)(nitems + 1));
  for (uint32_t i = 0; i < nitems; i++) for (int k = 0; k < 16; k++) sx[16 * (size_t)i + k] = (int32_t)((bsel[item[i]] >> (2 * k)) & 3);
  const uint32_t Ks_t = nitems < Kmax_s ? nitems : Kmax_s;
  uint32_t *sleaf = (uint32_t *)malloc(4 * (size_t)(nitems + 1)), Ks = 0;
  tsvq(sx, NULL, nitems, 16, WD16, Ks_t, sleaf, &Ks);
  uint32_t *scb = (uint32_t *)calloc(Ks + 1, 4);
  lstat *sst = (lstat *)malloc(sizeof(lstat) * (size_t)(Ks + 1));
  for (int it = 0; it < 2; it++) {
    memset(sst, 0, sizeof(lstat) * Ks);
    for (uint32_t i = 0; i < nitems; i++) { lstat *s = &sst[sleaf[i]]; s->W++; for (int k = 0; k < 16; k++) s->S[k] += sx[16 * (size_t)i + k]; }
    for (uint32_t k = 0; k < Ks; k++) { const lstat *s = &sst[k]; if (!s->W) continue; uint32_t v = 0; for (int d = 0; d < 16; d++) v |= (uint32_t)((2 * s->S[d] + s->W) / (2 * s->W)) << (2 * d); scb[k] = v; }
    for (uint32_t i = 0; i < nitems; i++) {
      const uint16_t (*et)[4] = etab[item[i]]; uint32_t bd = 0xffffffffu, bk = 0;
      for (uint32_t k = 0; k < Ks; k++) { uint32_t v = scb[k], d = 0; for (int t = 0; t < 16; t++) d += et[t][(v >> (2 * t)) & 3]; if (d < bd) { bd = d; bk = k; } }
      sleaf[i] = bk;
    }
  }
  uint32_t *scu = (uint32_t *)malloc(4 * (size_t)(Ks + 1)); uint32_t ns = 0;
  { uint8_t *used = (uint8_t *)calloc(Ks + 1, 1); for (uint32_t i = 0; i < nitems; i++) used[sleaf[i]] = 1; for (uint32_t k = 0; k < Ks; k++) if (used[k]) scu[ns++] = scb[k]; free(used); }
  qsort(scu, ns, 4, cmp_u32); ns = uniq_sorted(scu, ns);
  for (uint32_t i = 0; i < nitems; i++) bsi[item[i]] = (uint16_t)bsearch_u32(scu, ns, scb[sleaf[i]]);
  /* skipped blocks copy the previous layer's final indices */
  for (int l = stride; l < L; l++) for (uint32_t b = 0; b < nb; b++) if (skip[(size_t)l * nb + b]) { bei[(size_t)l * nb + b] = bei[(size_t)(l - stride) * nb + b]; bsi[(size_t)l * nb + b] = bsi[(size_t)(l - stride) * nb + b]; }

  /* ---- step I: symbolisation ---- */
  const uint32_t HS = 64, SEL_RLE = ns + HS;
  /* tokens: kind 0 endpoint_pred sym, 1 ep repeat (vlc4 extra), 2 delta endpoint, 3 selector sym, 4 selector rle sym (+vlc7 if 63) */
  typedef struct { uint8_t kind; uint16_t sym; uint32_t extra; } tok;
  tok **toks = (tok **)calloc((size_t)L, sizeof(tok *)); uint32_t *ntoks = (uint32_t *)calloc((size_t)L, 4);
  uint32_t *f_ep = (uint32_t *)calloc(257, 4), *f_de = (uint32_t *)calloc(ne + 1, 4), *f_sel = (uint32_t *)calloc(ns + HS + 2, 4), *f_rle = (uint32_t *)calloc(64, 4);
  uint8_t *pred = (uint8_t *)malloc(nb);
  for (int l = 0; l < L; l++) {
    const uint16_t *ei = bei + (size_t)l * nb, *si = bsi + (size_t)l * nb; const uint8_t *sk = skip + (size_t)l * nb; const int is_p = l >= stride;
    for (uint32_t y = 0; y < by; y++) for (uint32_t x = 0; x < bx; x++) {
      uint32_t b = y * bx + x; uint8_t p;
      if (is_p && sk[b]) p = 2;
      else if (x > 0 && ei[b] == ei[b - 1]) p = 0;
      else if (y > 0 && ei[b] == ei[b - bx]) p = 1;
      else if (!is_p && x > 0 && y > 0 && ei[b] == ei[b - bx - 1]) p = 2;
      else p = 3;
      pred[b] = p;
    }
    /* three fixed token slots per block, in bitstream order: [endpoint-pred] [delta endpoint] [selector];
       a run token sits in the slot of the run's first element, absorbed elements leave kind 255 (no bits) */
    const uint32_t nt = 3 * nb;
    tok *T = (tok *)malloc(sizeof(tok) * (size_t)(nt + 1));
    for (uint32_t i = 0; i < nt; i++) { T[i].kind = 255; T[i].sym = 0; T[i].extra = 0; }
    uint32_t histb[64]; for (uint32_t i = 0; i < HS; i++) histb[i] = i;
    uint32_t rover = HS / 2, prev_sym = 0, prev_ei = 0;
    uint32_t ep_count = 0, ep_s1 = 0, ep_s2 = 0, sel_count = 0, sel_s1 = 0, sel_s2 = 0;
#define FIN_EP() do { if (ep_count >= 3) { T[ep_s1].kind = 1; T[ep_s1].sym = 256; T[ep_s1].extra = ep_count - 3; f_ep[256]++; } \
      else { if (ep_count >= 1) { T[ep_s1].kind = 0; T[ep_s1].sym = (uint16_t)prev_sym; f_ep[prev_sym]++; } if (ep_count == 2) { T[ep_s2].kind = 0; T[ep_s2].sym = (uint16_t)prev_sym; f_ep[prev_sym]++; } } ep_count = 0; } while (0)
#define FIN_SEL() do { if (sel_count >= 3) { T[sel_s1].kind = 4; if (sel_count - 3 < 63) { T[sel_s1].sym = (uint16_t)(sel_count - 3); T[sel_s1].extra = 0; } else { T[sel_s1].sym = 63; T[sel_s1].extra = sel_count - 3; } f_sel[SEL_RLE]++; f_rle[T[sel_s1].sym]++; } \
      else { if (sel_count >= 1) { T[sel_s1].kind = 3; T[sel_s1].sym = (uint16_t)ns; f_sel[ns]++; } if (sel_count == 2) { T[sel_s2].kind = 3; T[sel_s2].sym = (uint16_t)ns; f_sel[ns]++; } } sel_count = 0; } while (0)
    for (uint32_t y = 0; y < by; y++) for (uint32_t x = 0; x < bx; x++) {
      const uint32_t b = y * bx + x;
      if (!(x & 1) && !(y & 1)) {
        uint32_t ms = pred[b];
        if (x + 1 < bx) ms |= (uint32_t)pred[b + 1] << 2;
        if (y + 1 < by) { ms |= (uint32_t)pred[b + bx] << 4; if (x + 1 < bx) ms |= (uint32_t)pred[b + bx + 1] << 6; }
        if (ms == prev_sym) { ep_count++; if (ep_count == 1) ep_s1 = 3 * b; else if (ep_count == 2) ep_s2 = 3 * b; }
        else { FIN_EP(); T[3 * b].kind = 0; T[3 * b].sym = (uint16_t)ms; f_ep[ms]++; prev_sym = ms; }
      }
      if (pred[b] == 3) {
        uint32_t d = ei[b] >= prev_ei ? ei[b] - prev_ei : ei[b] + ne - prev_ei;
        T[3 * b + 1].kind = 2; T[3 * b + 1].sym = (uint16_t)d; f_de[d]++;
      }
      prev_ei = ei[b];
      if (!(is_p && sk[b])) {
        uint32_t s = si[b], h = HS;
        for (uint32_t k = 0; k < HS; k++) if (histb[k] == s) { h = k; break; }
        if (h == 0) { sel_count++; if (sel_count == 1) sel_s1 = 3 * b + 2; else if (sel_count == 2) sel_s2 = 3 * b + 2; }
        else {
          FIN_SEL();
          if (h < HS) { T[3 * b + 2].kind = 3; T[3 * b + 2].sym = (uint16_t)(ns + h); f_sel[ns + h]++; uint32_t t_ = histb[h]; histb[h] = histb[h / 2]; histb[h / 2] = t_; }
          else { T[3 * b + 2].kind = 3; T[3 * b + 2].sym = (uint16_t)s; f_sel[s]++; histb[rover] = s; rover++; if (rover == HS) rover = HS / 2; }
        }
      }
    }
    FIN_SEL(); FIN_EP();
    toks[l] = T; ntoks[l] = nt;
  }
  /* the transcoder rejects empty models */
  { uint32_t s; s = 0; for (uint32_t i = 0; i < ne; i++) s |= f_de[i]; if (!s) f_de[0] = 1; s = 0; for (int i = 0; i < 64; i++) s |= f_rle[i]; if (!s) f_rle[0] = 1; }
  hcode h_ep, h_de, h_sel, h_rle;
  hcode_build(&h_ep, f_ep, 257, 16); hcode_build(&h_de, f_de, (int)ne, 16); hcode_build(&h_sel, f_sel, (int)(ns + HS + 1), 16); hcode_build(&h_rle, f_rle, 64, 16);
  /* ---- slices ---- */
  orc_buf level = {0}; uint32_t sl_off[KTX2_MAX_LAYERS], sl_len[KTX2_MAX_LAYERS];
  for (int l = 0; l < L; l++) {
    bitw w; memset(&w, 0, sizeof(w));
    for (uint32_t i = 0; i < ntoks[l]; i++) {
      const tok *t = &toks[l][i];
      switch (t->kind) {
        case 0: hcode_put(&w, &h_ep, t->sym); break;
        case 1: hcode_put(&w, &h_ep, 256); bw_vlc(&w, t->extra, 4); break;
        case 2: hcode_put(&w, &h_de, t->sym); break;
        case 3: hcode_put(&w, &h_sel, t->sym); break;
        case 4: hcode_put(&w, &h_sel, SEL_RLE); hcode_put(&w, &h_rle, t->sym); if (t->sym == 63) bw_vlc(&w, t->extra, 7); break;
        default: break;
      }
    }
    bw_flush(&w);
    sl_off[l] = (uint32_t)level.n; sl_len[l] = (uint32_t)w.b.n; ob_bytes(&level, w.b.p, w.b.n); ob_free(&w.b);
  }
  /* ---- codebooks ---- */
  bitw wep; memset(&wep, 0, sizeof(wep));
  { uint32_t fc[3][32], fi[8]; memset(fc, 0, sizeof(fc)); memset(fi, 0, sizeof(fi));
    int prev[3] = { 16, 16, 16 }, pi = 0;
    for (uint32_t k = 0; k < ne; k++) { int t = (int)(ecb[k] >> 15), c[3] = { (int)((ecb[k] >> 10) & 31), (int)((ecb[k] >> 5) & 31), (int)(ecb[k] & 31) };
      fi[(t - pi) & 7]++; pi = t;
      for (int d = 0; d < 3; d++) { int m = prev[d] <= 9 ? 0 : (prev[d] <= 21 ? 1 : 2); fc[m][(c[d] - prev[d]) & 31]++; prev[d] = c[d]; } }
    hcode hm[3], hi;
    for (int m = 0; m < 3; m++) { uint32_t s = 0; for (int i = 0; i < 32; i++) s |= fc[m][i]; if (!s) fc[m][0] = 1; hcode_build(&hm[m], fc[m], 32, 16); }
    hcode_build(&hi, fi, 8, 16);
    for (int m = 0; m < 3; m++) write_huff(&wep, &hm[m]);
    write_huff(&wep, &hi);
    bw_put(&wep, 0, 1);                                   /* not grayscale */
    prev[0] = prev[1] = prev[2] = 16; pi = 0;
    for (uint32_t k = 0; k < ne; k++) { int t = (int)(ecb[k] >> 15), c[3] = { (int)((ecb[k] >> 10) & 31), (int)((ecb[k] >> 5) & 31), (int)(ecb[k] & 31) };
      hcode_put(&wep, &hi, (uint32_t)((t - pi) & 7)); pi = t;
      for (int d = 0; d < 3; d++) { int m = prev[d] <= 9 ? 0 : (prev[d] <= 21 ? 1 : 2); hcode_put(&wep, &hm[m], (uint32_t)((c[d] - prev[d]) & 31)); prev[d] = c[d]; } }
    bw_flush(&wep);
    for (int m = 0; m < 3; m++) hcode_free(&hm[m]);
    hcode_free(&hi); }
  bitw wsel; memset(&wsel, 0, sizeof(wsel));
  { bw_put(&wsel, 0, 1); bw_put(&wsel, 0, 1); bw_put(&wsel, 0, 1);     /* global=0 hybrid=0 raw=0 */
    uint32_t fd[256]; memset(fd, 0, sizeof(fd));
    for (uint32_t k = 1; k < ns; k++) for (int j = 0; j < 4; j++) fd[((scu[k] >> (8 * j)) ^ (scu[k - 1] >> (8 * j))) & 255]++;
    { uint32_t s = 0; for (int i = 0; i < 256; i++) s |= fd[i]; if (!s) fd[0] = 1; }
    hcode hd; hcode_build(&hd, fd, 256, 16);
    write_huff(&wsel, &hd);
    for (int j = 0; j < 4; j++) bw_put(&wsel, (scu[0] >> (8 * j)) & 255, 8);
    for (uint32_t k = 1; k < ns; k++) for (int j = 0; j < 4; j++) hcode_put(&wsel, &hd, ((scu[k] >> (8 * j)) ^ (scu[k - 1] >> (8 * j))) & 255);
    bw_flush(&wsel); hcode_free(&hd); }
  bitw wtab; memset(&wtab, 0, sizeof(wtab));
  write_huff(&wtab, &h_ep); write_huff(&wtab, &h_de); write_huff(&wtab, &h_sel); write_huff(&wtab, &h_rle);
  bw_put(&wtab, HS, 13); bw_flush(&wtab);

  /* ---- KTX2 container ---- */
  static const uint8_t ident[12] = { 0xAB, 'K', 'T', 'X', ' ', '2', '0', 0xBB, '\r', '\n', 0x1A, '\n' };
  static const char writer[] = "uvol-mi355x etc1s 0.1";
  orc_buf kvd = {0};
  { ob_u32(&kvd, 12 + 12); ob_bytes(&kvd, "KTXanimData", 12); ob_u32(&kvd, 1); ob_u32(&kvd, 15); ob_u32(&kvd, 0);
    uint32_t wl = 10 + (uint32_t)sizeof(writer); ob_u32(&kvd, wl); ob_bytes(&kvd, "KTXwriter", 10); ob_bytes(&kvd, writer, sizeof(writer)); while (kvd.n & 3) ob_u8(&kvd, 0); }
  const uint32_t dfd_off = 80 + 24, dfd_len = has_alpha ? 60 : 44, kvd_off = dfd_off + dfd_len, kvd_len = (uint32_t)kvd.n;
  uint64_t sgd_off = (kvd_off + kvd_len + 7) & ~7ull;
  const uint64_t sgd_len = 20 + 20 * (uint64_t)Lr + wep.b.n + wsel.b.n + wtab.b.n;
  const uint64_t lvl_off = sgd_off + sgd_len;
  ob_bytes(out, ident, 12);
  ob_u32(out, 0); ob_u32(out, 1); ob_u32(out, W); ob_u32(out, H); ob_u32(out, 0); ob_u32(out, (uint32_t)Lr); ob_u32(out, 1); ob_u32(out, 1); ob_u32(out, 1);
  ob_u32(out, dfd_off); ob_u32(out, dfd_len); ob_u32(out, kvd_off); ob_u32(out, kvd_len); ob_u64(out, sgd_off); ob_u64(out, sgd_len);
  ob_u64(out, lvl_off); ob_u64(out, level.n); ob_u64(out, 0);
  /* DFD: ETC1S, BT709, sRGB */
  ob_u32(out, dfd_len); ob_u32(out, 0); ob_u16(out, 2); ob_u16(out, (uint16_t)(dfd_len - 4));
  ob_u8(out, 163); ob_u8(out, 1); ob_u8(out, 2); ob_u8(out, 0);
  ob_u8(out, 3); ob_u8(out, 3); ob_u8(out, 0); ob_u8(out, 0);
  for (int i = 0; i < 8; i++) ob_u8(out, 0);
  ob_u16(out, 0); ob_u8(out, 63); ob_u8(out, 0); ob_u8(out, 0); ob_u8(out, 0); ob_u8(out, 0); ob_u8(out, 0); ob_u32(out, 0); ob_u32(out, 0xFFFFFFFFu);
  if (has_alpha) {                                       /* second sample: channel 15 (AAA) at bit 64 */
    ob_u16(out, 64); ob_u8(out, 63); ob_u8(out, 15); ob_u8(out, 0); ob_u8(out, 0); ob_u8(out, 0); ob_u8(out, 0); ob_u32(out, 0); ob_u32(out, 0xFFFFFFFFu); }
  ob_bytes(out, kvd.p, kvd.n);
  while (out->n < sgd_off) ob_u8(out, 0);
  ob_u16(out, (uint16_t)ne); ob_u16(out, (uint16_t)ns); ob_u32(out, (uint32_t)wep.b.n); ob_u32(out, (uint32_t)wsel.b.n); ob_u32(out, (uint32_t)wtab.b.n); ob_u32(out, 0);
  for (int l = 0; l < Lr; l++) { ob_u32(out, l > 0 ? 2 : 0); ob_u32(out, sl_off[l * stride]); ob_u32(out, sl_len[l * stride]); ob_u32(out, has_alpha ? sl_off[l * stride + 1] : 0); ob_u32(out, has_alpha ? sl_len[l * stride + 1] : 0); }
  ob_bytes(out, wep.b.p, wep.b.n); ob_bytes(out, wsel.b.p, wsel.b.n); ob_bytes(out, wtab.b.p, wtab.b.n);
  ob_bytes(out, level.p, level.n);

  ob_free(&kvd); ob_free(&wep.b); ob_free(&wsel.b); ob_free(&wtab.b); ob_free(&level);
  hcode_free(&h_ep); hcode_free(&h_de); hcode_free(&h_sel); hcode_free(&h_rle);
  for (int l = 0; l < L; l++) free(toks[l]);
  free(toks); free(ntoks); free(f_ep); free(f_de); free(f_sel); free(f_rle); free(pred);
  free(B); free(skip); free(cell); free(hist); free(cid); free(cw); free(cx); free(cidx); free(cleaf); free(ent); free(est); free(ecb);
  free(bei); free(bsi); free(bsel); free(etab); free(item); free(sx); free(sleaf); free(scb); free(sst); free(scu);
  return 0;
}
