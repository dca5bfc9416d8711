/* oracle/ktx2_dec.c — TEST INFRASTRUCTURE (see oracle_common.h).
 * KTX2 container + BasisLZ ETC1S decoder: the conformance pin of the texture path.  Follows
 * SURVEY.md B.0–B.4 and the executed listings D.8/D.9 (run there against all 50 reference
 * .ktx2 fixtures).  Consumer it stands in for: src/lib/KTX2Loader.js:469-580 (basis_transcoder).
 */
#include "ktx2_oracle.h"
#include <stdio.h>

typedef struct { const uint8_t *b; size_t n; uint64_t pos; } bitr;
static inline uint32_t br_get(bitr *r, int nbits) {
  uint32_t v = 0;
  for (int i = 0; i < nbits; i++) { size_t by = (size_t)(r->pos >> 3); uint32_t bit = by < r->n ? (r->b[by] >> (r->pos & 7)) & 1 : 0; v |= bit << i; r->pos++; }
  return v;
}

/* canonical (deflate-style) Huffman decoder, max code length 16, codes matched MSB-first */
typedef struct { int n; uint8_t *sizes; uint32_t first_code[18], first_idx[18], count[18]; uint32_t *sorted; int valid; } huff;
static void huff_free(huff *h) { free(h->sizes); free(h->sorted); memset(h, 0, sizeof(*h)); }
static int huff_init(huff *h, const uint8_t *sizes, int n) {
  memset(h, 0, sizeof(*h));
  h->n = n; h->sizes = (uint8_t *)malloc((size_t)n + 1); memcpy(h->sizes, sizes, (size_t)n);
  for (int i = 0; i < n; i++) { if (sizes[i] > 16) return -1; if (sizes[i]) h->count[sizes[i]]++; }
  uint32_t code = 0, idx = 0;
  for (int l = 1; l <= 16; l++) { code = (code + h->count[l - 1]) << 1; h->first_code[l] = code; h->first_idx[l] = idx; idx += h->count[l]; }
  h->sorted = (uint32_t *)malloc(sizeof(uint32_t) * (idx + 1));
  uint32_t fill[18]; memcpy(fill, h->first_idx, sizeof(fill));
  for (int i = 0; i < n; i++) if (sizes[i]) h->sorted[fill[sizes[i]]++] = (uint32_t)i;
  h->valid = idx > 0;
  return 0;
}
static int huff_dec(const huff *h, bitr *r) {
  uint32_t code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | br_get(r, 1);
    if (h->count[l] && code >= h->first_code[l] && code - h->first_code[l] < h->count[l]) return (int)h->sorted[h->first_idx[l] + (code - h->first_code[l])];
  }
  return -1;
}
static const int ZZ[21] = { 17, 18, 19, 20, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15, 16 };
static int read_huff(bitr *r, huff *out) {
  memset(out, 0, sizeof(*out));
  uint32_t total = br_get(r, 14);
  if (total == 0) return 0;
  uint32_t ncl = br_get(r, 5); if (ncl < 1 || ncl > 21) return -1;
  uint8_t cls[21] = {0};
  for (uint32_t i = 0; i < ncl; i++) cls[ZZ[i]] = (uint8_t)br_get(r, 3);
  huff clt; if (huff_init(&clt, cls, 21)) return -1;
  uint8_t *sizes = (uint8_t *)calloc(total + 256, 1); uint32_t k = 0; int rc = 0;
  while (k < total) {
    int c = huff_dec(&clt, r); if (c < 0) { rc = -1; break; }
    if (c <= 16) sizes[k++] = (uint8_t)c;
    else if (c == 17) k += 3 + br_get(r, 3);
    else if (c == 18) k += 11 + br_get(r, 7);
    else { uint32_t rep = c == 19 ? 3 + br_get(r, 2) : 7 + br_get(r, 7); if (k == 0) { rc = -1; break; } uint8_t pv = sizes[k - 1]; for (uint32_t j = 0; j < rep && k < total + 200; j++) sizes[k++] = pv; }
  }
  if (!rc && k != total) rc = -2;
  if (!rc) rc = huff_init(out, sizes, (int)total);
  free(sizes); huff_free(&clt);
  return rc;
}
static uint32_t vlc(bitr *r, int cb) {
  uint32_t v = 0; int ofs = 0;
  for (;;) { uint32_t s = br_get(r, cb + 1); v |= (s & ((1u << cb) - 1)) << ofs; ofs += cb; if (!(s >> cb) || ofs > 28) break; }
  return v;
}

static uint32_t rd32(const uint8_t *b) { uint32_t v; memcpy(&v, b, 4); return v; }
static uint64_t rd64(const uint8_t *b) { uint64_t v; memcpy(&v, b, 8); return v; }

void ktx2_free(ktx2_file *f) { free(f->endpoints); free(f->selectors); free(f->block_ei); free(f->block_si); memset(f, 0, sizeof(*f)); }

int ktx2_decode(const uint8_t *b, size_t n, ktx2_file *f) {
  static const uint8_t ident[12] = { 0xAB, 'K', 'T', 'X', ' ', '2', '0', 0xBB, '\r', '\n', 0x1A, '\n' };
  memset(f, 0, sizeof(*f));
  if (n < 104 || memcmp(b, ident, 12)) return -1;
  f->vk_format = rd32(b + 12); f->type_size = rd32(b + 16); f->width = rd32(b + 20); f->height = rd32(b + 24); f->depth = rd32(b + 28);
  f->layers = rd32(b + 32); f->faces = rd32(b + 36); f->levels = rd32(b + 40); f->supercomp = rd32(b + 44);
  f->dfd_off = rd32(b + 48); f->dfd_len = rd32(b + 52); f->kvd_off = rd32(b + 56); f->kvd_len = rd32(b + 60);
  f->sgd_off = rd64(b + 64); f->sgd_len = rd64(b + 72);
  f->level_off = rd64(b + 80); f->level_len = rd64(b + 88); f->level_ulen = rd64(b + 96);
  if (f->vk_format != 0 || f->supercomp != 1 || f->levels != 1 || f->faces != 1) return -2;
  if (f->dfd_off + f->dfd_len > n || f->kvd_off + f->kvd_len > n || f->sgd_off + f->sgd_len > n || f->level_off + f->level_len > n) return -3;
  if (f->dfd_len >= 44) { const uint8_t *d = b + f->dfd_off; f->dfd_model = d[12]; f->dfd_primaries = d[13]; f->dfd_transfer = d[14]; }
  /* key/value data */
  { size_t o = f->kvd_off, e = f->kvd_off + f->kvd_len;
    while (o + 4 <= e) {
      uint32_t len = rd32(b + o); o += 4; if (o + len > e) break;
      const char *key = (const char *)(b + o); size_t kl = strnlen(key, len);
      if (!strcmp(key, "KTXwriter")) { size_t vl = len - kl - 1; if (vl > 63) vl = 63; memcpy(f->writer, b + o + kl + 1, vl); f->writer[vl] = 0; }
      if (!strcmp(key, "KTXanimData") && len >= kl + 1 + 12) { f->has_anim = 1; f->anim_duration = rd32(b + o + kl + 1); f->anim_timescale = rd32(b + o + kl + 5); f->anim_loops = rd32(b + o + kl + 9); }
      o += (len + 3) & ~3u;
    } }
  const int nimg = (int)(f->layers ? f->layers : 1);
  if (nimg > KTX2_MAX_LAYERS) return -4;
  const uint8_t *s = b + f->sgd_off;
  if (f->sgd_len < 20 + 20 * (uint64_t)nimg) return -5;
  /* alpha slices: announced by a second DFD sample (channel 15, AAA) and carried by the image descs' second offset / length pair;
   * all images of a file have one or none.  Slice order: rgb0 a0 rgb1 a1 ...; a P-frame slice refers to the previous slice of its kind */
  { int any = 0, all = 1; for (int i = 0; i < nimg; i++) { const uint8_t *d = s + 20 + 20 * i; if (rd32(d + 16)) any = 1; else all = 0; }
    if (any != all) return -6;
    f->has_alpha = any;
    if (any && !(f->dfd_len >= 60 && (b[f->dfd_off + 44 + 3] & 15) == 15)) return -6; }
  const int stride = f->has_alpha ? 2 : 1, nsl = nimg * stride;
  if (nsl > KTX2_MAX_LAYERS) return -4;
  f->n_slices = nsl;
  f->n_endpoints = s[0] | (s[1] << 8); f->n_selectors = s[2] | (s[3] << 8);
  f->endpoints_len = rd32(s + 4); f->selectors_len = rd32(s + 8); f->tables_len = rd32(s + 12); f->extended_len = rd32(s + 16);
  for (int i = 0; i < nimg; i++) { const uint8_t *d = s + 20 + 20 * i;
    for (int k = 0; k < stride; k++) { f->slice_flags[i * stride + k] = rd32(d); f->slice_off[i * stride + k] = rd32(d + 4 + 8 * k); f->slice_len[i * stride + k] = rd32(d + 8 + 8 * k); } }
  const uint8_t *p = s + 20 + 20 * nimg;
  if ((uint64_t)(p - s) + f->endpoints_len + f->selectors_len + f->tables_len > f->sgd_len) return -5;
  const uint32_t ne = f->n_endpoints, ns = f->n_selectors;
  int rc = 0;
  /* ---- endpoint codebook (B.2) ---- */
  { bitr R = { p, f->endpoints_len, 0 }; huff m[3], mi;
    if (read_huff(&R, &m[0]) || read_huff(&R, &m[1]) || read_huff(&R, &m[2]) || read_huff(&R, &mi)) return -7;
    int gray = (int)br_get(&R, 1);
    f->endpoints = (uint8_t *)malloc(4 * (size_t)ne + 4);
    int prev[3] = { 16, 16, 16 }, pi = 0;
    for (uint32_t i = 0; i < ne && !rc; i++) {
      int d = huff_dec(&mi, &R); if (d < 0) { rc = -7; break; } pi = (d + pi) & 7;
      for (int c = 0; c < (gray ? 1 : 3); c++) {
        const huff *mm = prev[c] <= 9 ? &m[0] : (prev[c] <= 21 ? &m[1] : &m[2]);
        int dd = huff_dec(mm, &R); if (dd < 0) { rc = -7; break; }
        prev[c] = (prev[c] + dd) & 31;
      }
      if (gray) prev[1] = prev[2] = prev[0];
      f->endpoints[4 * i] = (uint8_t)prev[0]; f->endpoints[4 * i + 1] = (uint8_t)prev[1]; f->endpoints[4 * i + 2] = (uint8_t)prev[2]; f->endpoints[4 * i + 3] = (uint8_t)pi;
    }
    f->ep_bits_used = (uint32_t)R.pos;
    for (int k = 0; k < 3; k++) huff_free(&m[k]);
    huff_free(&mi);
    if (rc) return rc; }
  /* ---- selector codebook ---- */
  { bitr R = { p + f->endpoints_len, f->selectors_len, 0 };
    int global = (int)br_get(&R, 1), hybrid = (int)br_get(&R, 1), raw = (int)br_get(&R, 1);
    if (global || hybrid) return -8;
    f->selectors = (uint32_t *)malloc(4 * (size_t)ns + 4);
    if (raw) { for (uint32_t i = 0; i < ns; i++) { uint32_t v = 0; for (int j = 0; j < 4; j++) v |= br_get(&R, 8) << (8 * j); f->selectors[i] = v; } }
    else {
      huff m; if (read_huff(&R, &m)) return -8;
      uint8_t prevb[4] = {0, 0, 0, 0};
      for (uint32_t i = 0; i < ns && !rc; i++) {
        uint32_t v = 0;
        for (int j = 0; j < 4; j++) {
          uint8_t cur;
          if (i == 0) cur = (uint8_t)br_get(&R, 8);
          else { int d = huff_dec(&m, &R); if (d < 0) { rc = -8; break; } cur = (uint8_t)(d ^ prevb[j]); }
          prevb[j] = cur; v |= (uint32_t)cur << (8 * j);
        }
        f->selectors[i] = v;
      }
      huff_free(&m);
      if (rc) return rc;
    }
    f->sel_bits_used = (uint32_t)R.pos; }
  /* ---- tables ---- */
  huff epm, dem, sm, rle;
  { bitr R = { p + f->endpoints_len + f->selectors_len, f->tables_len, 0 };
    if (read_huff(&R, &epm) || read_huff(&R, &dem) || read_huff(&R, &sm) || read_huff(&R, &rle)) return -9;
    f->hist_size = br_get(&R, 13);
    f->tab_bits_used = (uint32_t)R.pos; }
  if (!epm.valid || !dem.valid || !sm.valid || !rle.valid) { rc = -9; goto done; }
  /* ---- slices (B.3) ---- */
  {
    const uint32_t bx = (f->width + 3) / 4, by = (f->height + 3) / 4, hs = f->hist_size;
    f->bx = bx; f->by = by;
    if (hs == 0 || hs > 64) { rc = -10; goto done; }
    f->block_ei = (uint16_t *)malloc(2 * (size_t)bx * by * nsl); f->block_si = (uint16_t *)malloc(2 * (size_t)bx * by * nsl);
    uint16_t *pe[2]; uint8_t *pb[2];
    pe[0] = (uint16_t *)calloc(bx + 1, 2); pe[1] = (uint16_t *)calloc(bx + 1, 2); pb[0] = (uint8_t *)calloc(bx + 1, 1); pb[1] = (uint8_t *)calloc(bx + 1, 1);
    for (int sl = 0; sl < nsl && !rc; sl++) {
      if ((uint64_t)f->slice_off[sl] + f->slice_len[sl] > f->level_len) { rc = -11; break; }
      const int is_p = (f->slice_flags[sl] & 2) != 0;
      if (is_p && sl < stride) { rc = -11; break; }
      bitr R = { b + f->level_off + f->slice_off[sl], f->slice_len[sl], 0 };
      uint16_t *oe = f->block_ei + (size_t)sl * bx * by, *os = f->block_si + (size_t)sl * bx * by;
      const uint16_t *pve = sl >= stride ? oe - (size_t)stride * bx * by : NULL, *pvs = sl >= stride ? os - (size_t)stride * bx * by : NULL;
      uint32_t hist[64]; for (uint32_t i = 0; i < hs; i++) hist[i] = i;
      uint32_t rover = hs / 2, RLE = ns + hs, prev_sym = 0, rep = 0, prev_ei = 0, sel_rle = 0, nskip = 0;
      memset(pe[0], 0, 2 * (bx + 1)); memset(pe[1], 0, 2 * (bx + 1)); memset(pb[0], 0, bx + 1); memset(pb[1], 0, bx + 1);
      for (uint32_t y = 0; y < by && !rc; y++) {
        const int cur = y & 1;
        uint32_t pbits = 0;
        for (uint32_t x = 0; x < bx; x++) {
          if ((x & 1) == 0) {
            if ((y & 1) == 0) {
              if (rep) { rep--; pbits = prev_sym; }
              else {
                int d = huff_dec(&epm, &R); if (d < 0) { rc = -12; break; }
                if (d == 256) { rep = vlc(&R, 4) + 3 - 1; pbits = prev_sym; } else { prev_sym = (uint32_t)d; pbits = (uint32_t)d; }
              }
              pb[cur ^ 1][x] = (uint8_t)(pbits >> 4);
            } else pbits = pb[cur][x];
          }
          const uint32_t pred = pbits & 3; pbits >>= 2;
          uint32_t ei; int skip = 0;
          if (pred == 0) { if (x == 0) { rc = -13; break; } ei = prev_ei; }
          else if (pred == 1) { if (y == 0) { rc = -13; break; } ei = pe[cur ^ 1][x]; }
          else if (pred == 2) {
            if (is_p) { skip = 1; ei = pve[x + y * bx]; }
            else { if (x == 0 || y == 0) { rc = -13; break; } ei = pe[cur ^ 1][x - 1]; }
          } else { int d = huff_dec(&dem, &R); if (d < 0) { rc = -12; break; } ei = (uint32_t)d + prev_ei; if (ei >= ne) ei -= ne; }
          if (ei >= ne) { rc = -14; break; }
          pe[cur][x] = (uint16_t)ei; prev_ei = ei;
          uint32_t si;
          if (skip) { si = pvs[x + y * bx]; nskip++; }
          else {
            uint32_t sym;
            if (sel_rle > 0) { sel_rle--; sym = ns; }
            else {
              int d = huff_dec(&sm, &R); if (d < 0) { rc = -12; break; } sym = (uint32_t)d;
              if (sym == RLE) { int rr = huff_dec(&rle, &R); if (rr < 0) { rc = -12; break; } sel_rle = rr == 63 ? vlc(&R, 7) + 3 : (uint32_t)rr + 3; sym = ns; sel_rle--; }
            }
            if (sym >= ns) { uint32_t h = sym - ns; if (h >= hs) { rc = -14; break; } si = hist[h]; if (h) { uint32_t t = hist[h]; hist[h] = hist[h / 2]; hist[h / 2] = t; } }
            else { si = sym; hist[rover] = si; rover++; if (rover == hs) rover = hs / 2; }
          }
          if (si >= ns) { rc = -14; break; }
          oe[x + y * bx] = (uint16_t)ei; os[x + y * bx] = (uint16_t)si;
        }
      }
      f->slice_bits_used[sl] = R.pos; f->slice_skip[sl] = nskip;
      if (!rc && R.pos > 8ull * f->slice_len[sl]) rc = -15;
    }
    free(pe[0]); free(pe[1]); free(pb[0]); free(pb[1]);
  }
done:
  huff_free(&epm); huff_free(&dem); huff_free(&sm); huff_free(&rle);
  if (rc) ktx2_free(f);
  return rc;
}

static const int INTEN[8][4] = { {-8, -2, 2, 8}, {-17, -5, 5, 17}, {-29, -9, 9, 29}, {-42, -13, 13, 42}, {-60, -18, 18, 60}, {-80, -24, 24, 80}, {-106, -33, 33, 106}, {-183, -47, 47, 183} };
static inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void ktx2_layer_rgba(const ktx2_file *f, int layer, uint8_t *out) {
  const uint32_t bx = f->bx, by = f->by, W = f->width, H = f->height;
  const int stride = f->has_alpha ? 2 : 1;
  const uint16_t *ei = f->block_ei + (size_t)layer * stride * bx * by, *si = f->block_si + (size_t)layer * stride * bx * by;
  const uint16_t *aei = ei + (size_t)bx * by, *asi = si + (size_t)bx * by;          /* the alpha slice (read only with has_alpha) */
  for (uint32_t Y = 0; Y < by; Y++) for (uint32_t X = 0; X < bx; X++) {
    const uint8_t *e = f->endpoints + 4 * (size_t)ei[X + Y * bx]; const uint32_t sel = f->selectors[si[X + Y * bx]];
    int base[3]; for (int c = 0; c < 3; c++) base[c] = (e[c] << 3) | (e[c] >> 2);
    const uint8_t *ae = f->has_alpha ? f->endpoints + 4 * (size_t)aei[X + Y * bx] : NULL; const uint32_t asel = f->has_alpha ? f->selectors[asi[X + Y * bx]] : 0;
    const int abase = ae ? ((ae[1] << 3) | (ae[1] >> 2)) : 0;
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
      uint32_t px = X * 4 + x, py = Y * 4 + y; if (px >= W || py >= H) continue;
      int s = (int)((sel >> (8 * y + 2 * x)) & 3), d = INTEN[e[3]][s];
      uint8_t *o = out + 4 * ((size_t)py * W + px);
      o[0] = clamp255(base[0] + d); o[1] = clamp255(base[1] + d); o[2] = clamp255(base[2] + d);
      o[3] = ae ? clamp255(abase + INTEN[ae[3]][(asel >> (8 * y + 2 * x)) & 3]) : 255;
    }
  }
}
