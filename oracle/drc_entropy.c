/* oracle/drc_entropy.c — TEST INFRASTRUCTURE (see oracle_common.h).
 * rANS multi-symbol ("raw" scheme) and rabs binary coders of the Draco bitstream.
 * Decode side follows SURVEY.md A.2 / D.1 / D.2 (executed there against all 250 reference
 * fixtures); encode side follows SURVEY.md A.10 / D.7 (byte-identical on fixture sections).
 * The third-party origin is google/draco (decoder 1.4.3, src/V2/player.ts:101), not vendored.
 */
#include "drc_oracle.h"
#include <math.h>

static const uint32_t CRC_POLY = 0xEDB88320u;
uint32_t orc_crc32(const void *data, size_t n) {
  static uint32_t tab[256]; static int init = 0;
  if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? (CRC_POLY ^ (c >> 1)) : (c >> 1); tab[i] = c; } init = 1; }
  uint32_t c = 0xFFFFFFFFu; const uint8_t *p = (const uint8_t *)data;
  for (size_t i = 0; i < n; i++) c = tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

static int rd_varint(const uint8_t *b, size_t n, size_t *o, uint64_t *v) {
  uint64_t r = 0; int s = 0;
  for (;;) {
    if (*o >= n || s > 63) return -1;
    uint8_t c = b[(*o)++]; r |= (uint64_t)(c & 0x7f) << s; s += 7;
    if (c < 0x80) break;
  }
  *v = r; return 0;
}

/* tagged tail read shared by rANS and rabs */
static int ans_read_init(const uint8_t *buf, size_t n, size_t *off, uint32_t *st, uint32_t L, int allow3) {
  if (n == 0) return -1;
  int x = buf[n - 1] >> 6; *off = n;
  if (x == 0) { *st = buf[n - 1] & 0x3f; *off -= 1; }
  else if (x == 1) { if (n < 2) return -1; *st = ((uint32_t)buf[n - 2] | (uint32_t)buf[n - 1] << 8) & 0x3fff; *off -= 2; }
  else if (x == 2) { if (n < 3) return -1; *st = ((uint32_t)buf[n - 3] | (uint32_t)buf[n - 2] << 8 | (uint32_t)buf[n - 1] << 16) & 0x3fffff; *off -= 3; }
  else { if (!allow3 || n < 4) return -1; *st = ((uint32_t)buf[n - 4] | (uint32_t)buf[n - 3] << 8 | (uint32_t)buf[n - 2] << 16 | (uint32_t)buf[n - 1] << 24) & 0x3fffffff; *off -= 4; }
  *st += L; return 0;
}

/* rANS table + payload of one RAnsSymbolDecoder<bl>: the RAW scheme's symbols, or the TAGGED scheme's bit-length tags (bl = 5) */
static int rans_section(const uint8_t *b, size_t n, size_t *o, int bl, uint32_t nvals, uint32_t *out, orc_sym_info *info, int scheme);

int orc_decode_symbols(const uint8_t *b, size_t n, size_t *o, uint32_t nvals, uint32_t *out, orc_sym_info *info) { return orc_decode_symbols_nc(b, n, o, nvals, 1, out, info); }

int orc_decode_symbols_nc(const uint8_t *b, size_t n, size_t *o, uint32_t nvals, int ncomp, uint32_t *out, orc_sym_info *info) {
  if (*o + 1 > n) return -1;
  int scheme = b[(*o)++];
  if (scheme == 0) {
    /* TAGGED scheme (DecodeTaggedSymbols; no reference fixture uses it - restated from the published bitstream description, pinned only by
     * the round trip through this file's own writer in the tests): one rANS-coded tag per group of ncomp values = their bit length
     * (alphabet 0..32, RAnsSymbolDecoder<5>: 12-bit precision), then the values as raw LSB-first bit fields of that length */
    if (ncomp < 1) return -1;
    const uint32_t ntags = nvals / (uint32_t)ncomp;
    uint32_t *tags = (uint32_t *)malloc(4 * ((size_t)ntags + 1));
    int rc = rans_section(b, n, o, 5, ntags, tags, info, 0);
    if (!rc) {
      size_t bit = 0; const uint8_t *p = b + *o; const size_t avail = (n - *o) * 8;
      for (uint32_t t = 0; t < ntags && !rc; t++) {
        const uint32_t len = tags[t]; if (len > 32) { rc = -7; break; }
        for (int c = 0; c < ncomp; c++) {
          uint32_t v = 0;
          if (bit + len > avail) { rc = -1; break; }
          for (uint32_t k = 0; k < len; k++, bit++) v |= (uint32_t)((p[bit >> 3] >> (bit & 7)) & 1) << k;
          out[(size_t)t * ncomp + c] = v;
        }
      }
      *o += (bit + 7) / 8;
    }
    free(tags);
    return rc;
  }
  if (scheme != 1) return -2;
  if (*o + 1 > n) return -1;
  int bl = b[(*o)++];
  return rans_section(b, n, o, bl, nvals, out, info, 1);
}

static int rans_section(const uint8_t *b, size_t n, size_t *o, int bl, uint32_t nvals, uint32_t *out, orc_sym_info *info, int scheme) {
  int prec_bits = (3 * bl) / 2; if (prec_bits < 12) prec_bits = 12; if (prec_bits > 20) prec_bits = 20;
  uint32_t prec = 1u << prec_bits, L = prec * 4;
  uint64_t ns; if (rd_varint(b, n, o, &ns)) return -1;
  if (ns > (1u << 20)) return -3;
  uint32_t *probs = (uint32_t *)calloc(ns ? ns : 1, 4), *cum = (uint32_t *)calloc(ns ? ns : 1, 4);
  uint32_t *lut = (uint32_t *)malloc((size_t)prec * 4);
  int rc = 0; uint32_t i = 0; int unique = 0;
  while (i < ns) {
    if (*o >= n) { rc = -1; goto done; }
    uint8_t pd = b[(*o)++]; int tok = pd & 3;
    if (tok == 3) { uint32_t run = (pd >> 2) + 1; if (i + run > ns) { rc = -4; goto done; } i += run; }
    else { uint32_t p = pd >> 2; for (int k = 0; k < tok; k++) { if (*o >= n) { rc = -1; goto done; } p |= (uint32_t)b[(*o)++] << (8 * (k + 1) - 2); } probs[i++] = p; }
  }
  { uint64_t c = 0;
    for (i = 0; i < ns; i++) { cum[i] = (uint32_t)c; if (probs[i]) unique++; if (c + probs[i] > prec) { rc = -5; goto done; } for (uint32_t j = 0; j < probs[i]; j++) lut[c + j] = i; c += probs[i]; }
    if (c != prec && nvals > 0) { rc = -5; goto done; } }
  { uint64_t len; if (rd_varint(b, n, o, &len)) { rc = -1; goto done; }
    if (*o + len > n) { rc = -1; goto done; }
    const uint8_t *buf = b + *o; size_t off; uint32_t st;
    if (ans_read_init(buf, (size_t)len, &off, &st, L, 1)) { rc = -6; goto done; }
    for (uint32_t k = 0; k < nvals; k++) {
      while (st < L && off > 0) { off--; st = st * 256 + buf[off]; }
      uint32_t quo = st / prec, rem = st % prec, s = lut[rem];
      st = quo * probs[s] + rem - cum[s];
      out[k] = s;
    }
    while (st < L && off > 0) { off--; st = st * 256 + buf[off]; }   /* final renormalisation pull (A.2 invariant) */
    if (info) { info->scheme = scheme; info->bl = bl; info->prec_bits = prec_bits; info->alphabet = (int)ns; info->unique = unique;
                info->left = (int)off; info->final_state = st; info->base = L; info->payload = (size_t)len; }
    *o += len; }
done:
  free(probs); free(cum); free(lut);
  return rc;
}

/* ---- encoder: probability table (D.7 build_table), serialisation, payload ---- */
typedef struct { uint32_t p; uint32_t id; } prob_ent;
static int cmp_prob(const void *a, const void *b) {
  const prob_ent *x = (const prob_ent *)a, *y = (const prob_ent *)b;
  if (x->p != y->p) return x->p < y->p ? -1 : 1;
  return x->id < y->id ? -1 : (x->id > y->id);     /* stable ascending */
}

static int msb32(uint32_t v) { int r = -1; while (v) { r++; v >>= 1; } return r; }

void orc_encode_symbols(const uint32_t *syms, uint32_t nvals, orc_buf *out) {
  uint32_t maxv = 0;
  for (uint32_t i = 0; i < nvals; i++) if (syms[i] > maxv) maxv = syms[i];
  uint32_t ns = maxv + 1;
  uint64_t *freq = (uint64_t *)calloc(ns, 8);
  for (uint32_t i = 0; i < nvals; i++) freq[syms[i]]++;
  uint32_t uniq = 0; for (uint32_t i = 0; i < ns; i++) if (freq[i]) uniq++;
  int bl = msb32(uniq) + 1; if (bl < 1) bl = 1; if (bl > 18) bl = 18;   /* compression level 7: no adjustment */
  int prec_bits = (3 * bl) / 2; if (prec_bits < 12) prec_bits = 12; if (prec_bits > 20) prec_bits = 20;
  uint32_t prec = 1u << prec_bits, L = prec * 4;
  uint32_t *probs = (uint32_t *)calloc(ns, 4), *cum = (uint32_t *)calloc(ns, 4);
  uint64_t tot = 0; double total = (double)nvals;
  for (uint32_t i = 0; i < ns; i++) if (freq[i]) {
    uint32_t p = (uint32_t)(((double)freq[i] / total) * (double)prec + 0.5);
    if (p == 0) p = 1;
    probs[i] = p; tot += p;
  }
  if (tot != prec) {
    prob_ent *ord = (prob_ent *)malloc(sizeof(prob_ent) * ns);
    for (uint32_t i = 0; i < ns; i++) { ord[i].p = probs[i]; ord[i].id = i; }
    qsort(ord, ns, sizeof(prob_ent), cmp_prob);
    if (tot < prec) probs[ord[ns - 1].id] += (uint32_t)(prec - tot);
    else {
      int64_t err = (int64_t)tot - prec;
      while (err > 0) {
        double rel = (double)prec / (double)tot;
        for (int64_t j = (int64_t)ns - 1; j > 0; j--) {
          uint32_t sid = ord[j].id;
          if (probs[sid] <= 1) { if (j == (int64_t)ns - 1) { err = 0; } break; }
          int32_t newp = (int32_t)floor(rel * (double)probs[sid]);
          int32_t fix = (int32_t)probs[sid] - newp;
          if (fix == 0) fix = 1;
          if (fix >= (int32_t)probs[sid]) fix = (int32_t)probs[sid] - 1;
          if (fix > err) fix = (int32_t)err;
          probs[sid] -= fix; tot -= fix; err -= fix;
          if (tot == prec) break;
        }
      }
    }
    free(ord);
  }
  { uint32_t c = 0; for (uint32_t i = 0; i < ns; i++) { cum[i] = c; c += probs[i]; } }
  ob_u8(out, 1); ob_u8(out, (uint8_t)bl); ob_varint(out, ns);
  for (uint32_t i = 0; i < ns;) {
    uint32_t p = probs[i];
    if (p == 0) {
      uint32_t off = 0;
      while (off < 63 && i + off + 1 < ns && probs[i + off + 1] == 0) off++;
      ob_u8(out, (uint8_t)((off << 2) | 3)); i += off + 1;
    } else {
      int nb = p < (1u << 6) ? 0 : (p < (1u << 14) ? 1 : 2);
      ob_u8(out, (uint8_t)(((p << 2) | nb) & 0xff));
      for (int k = 0; k < nb; k++) ob_u8(out, (uint8_t)((p >> (8 * (k + 1) - 2)) & 0xff));
      i++;
    }
  }
  orc_buf pl = {0};
  uint32_t st = L;
  for (int64_t i = (int64_t)nvals - 1; i >= 0; i--) {
    uint32_t s = syms[i], p = probs[s];
    uint64_t lim = (uint64_t)(L / prec) * 256 * p;
    while (st >= lim) { ob_u8(&pl, (uint8_t)(st & 255)); st >>= 8; }
    st = (st / p) * prec + st % p + cum[s];
  }
  st -= L;
  if (st < (1u << 6)) ob_u8(&pl, (uint8_t)st);
  else if (st < (1u << 14)) { uint32_t v = (1u << 14) + st; ob_u8(&pl, v & 255); ob_u8(&pl, (v >> 8) & 255); }
  else if (st < (1u << 22)) { uint32_t v = (2u << 22) + st; ob_u8(&pl, v & 255); ob_u8(&pl, (v >> 8) & 255); ob_u8(&pl, (v >> 16) & 255); }
  else { uint32_t v = (3u << 30) + st; ob_u32(&pl, v); }
  ob_varint(out, pl.n); ob_bytes(out, pl.p, pl.n);
  ob_free(&pl); free(freq); free(probs); free(cum);
}

int orc_rabs_open(orc_rabs_dec *r, const uint8_t *b, size_t n, size_t o) {
  if (o >= n) return -1;
  r->p0 = b[o++]; uint64_t len; if (rd_varint(b, n, &o, &len)) return -1;
  if (o + len > n) return -1;
  r->buf = b + o; r->end = o + (size_t)len;
  if (len == 0) { r->st = 4096; r->off = 0; return 0; }
  return ans_read_init(r->buf, (size_t)len, &r->off, &r->st, 4096, 0);
}
int orc_rabs_bit(orc_rabs_dec *r) {
  uint32_t p = 256u - r->p0;
  if (r->st < 4096 && r->off > 0) { r->off--; r->st = r->st * 256 + r->buf[r->off]; }
  uint32_t quot = r->st / 256, rem = r->st % 256, xn = quot * p;
  if (rem < p) { r->st = xn + rem; return 1; }
  r->st = r->st - xn - p; return 0;
}
void orc_rabs_encode(const uint8_t *bits, size_t nbits, orc_buf *out) {
  uint64_t zeros = 0; for (size_t i = 0; i < nbits; i++) zeros += !bits[i];
  uint64_t total = nbits ? nbits : 1;
  uint32_t p0raw = (uint32_t)(((double)zeros / (double)total) * 256.0 + 0.5);
  uint32_t p0 = p0raw < 255 ? p0raw : 255; if (p0 == 0) p0 = 1;
  uint32_t p = 256 - p0, st = 4096;
  orc_buf pl = {0};
  for (int64_t i = (int64_t)nbits - 1; i >= 0; i--) {
    int bit = bits[i]; uint32_t ls = bit ? p : p0;
    if (st >= 16u * 256u * ls) { ob_u8(&pl, (uint8_t)(st & 255)); st >>= 8; }
    uint32_t q = st / ls, rr = st % ls;
    st = q * 256 + rr + (bit ? 0 : p);
  }
  st -= 4096;
  if (st < (1u << 6)) ob_u8(&pl, (uint8_t)st);
  else if (st < (1u << 14)) { uint32_t v = (1u << 14) + st; ob_u8(&pl, v & 255); ob_u8(&pl, (v >> 8) & 255); }
  else { uint32_t v = (2u << 22) + st; ob_u8(&pl, v & 255); ob_u8(&pl, (v >> 8) & 255); ob_u8(&pl, (v >> 16) & 255); }
  ob_u8(out, (uint8_t)p0); ob_varint(out, pl.n); ob_bytes(out, pl.p, pl.n);
  ob_free(&pl);
}
