// geom_decode.hip — batched Draco 2.2 mesh decode (geometry half): `.drc` (TRIANGULAR_MESH, valence
// edgebreaker, the attribute decoders of SURVEY A.0) -> per-attribute value arrays + per-corner indices.
// Replaces what the stock player obtains from the draco WASM decoder per frame (reference src/V2/player.ts:101, :313-336:
// DRACOLoader -> BufferGeometry with position / uv / normal).  Bitstream: SURVEY Appendix A.1-A.9.
//
// Stages (n frames per call, one launch per stage; serial stages run one lane per frame / stream / attribute):
//   k_gdec_index    1 lane / frame     header, split events, every section's offset (rabs streams, symbol tables, payloads)
//   k_gdec_rans     1 wave / stream    probability tables (lanes), slot->symbol LUT, serial rANS symbol decode
//   k_gdec_conn     1 lane / frame     connectivity state machine (C/S/L/R/E + valence contexts + start faces) -> opp, c2v
//   k_gdec_seams    1 lane / frame     attribute seam bits (rabs) in corner order
//   k_gdec_atttab   1 lane / (attribute, frame)  attribute corner tables (vertex ids per seam-separated fan segment)
//   k_gdec_open     parallel           on-boundary flags, job scalars for the shared traversal kernels
//   k_pack_faces / k_traverse / k_v2d  (geom_encode.hip) the SAME DepthFirstTraverser kernels the encoder uses
//   k_gdec_rans     again for the attribute symbol streams (their lengths are the traversal results)
//   k_gdec_pred     1 lane / (decoder, frame)   parallelogram / tex-coord-portable / geometric-normal prediction + transforms
//   k_gdec_finish   parallel           dequantisation (fp32, octahedral) and per-corner entry indices
#include "uvol_common.hpp"
#include "geom_device.hpp"
#include "uvol_ws.hpp"

#define GD_NRS 10          // rANS streams: 0..5 valence contexts, 6 + d = attribute decoder d (d < 4)
#define GD_MAXDEC 4
#define GD_MAXAD 4
#define GD_MAX_NS (1u << 18)   // largest symbol alphabet of an attribute stream (16-bit quantisation residuals fit)
#define GD_CTX_NS 16
#define GD_E_WS_OVERFLOW (-50)      // the compact workspace cannot hold this frame (more entries per face than usual): retried with worst-case sizes

struct GDRans { uint32_t present, scheme, bl, prec_bits, ns, max_ns, max_prec_bits, tab_off, pay_off, pay_len, nvals; uint32_t *probs, *cum, *lut, *out;
                uint32_t early, redo; };      // attribute streams: decoded beside the traversal with the count the tables predict / decoded again after it (count differed)
struct GDRabs { uint32_t present, p0, pay_off, pay_len; };
struct GDAtt {
  int32_t att_data_id, dec_type, att_type, data_type, ncomp, unique_id, seq_type, pred_method, transform, nc;
  int32_t lo, hi, n_orient, maxq, cen, qbits; float minv[4], range;
  int32_t table;                 // 0 = base corner table, 1 + i = attribute table i
  int32_t *vals;                 // ne * nc decoded integers, entry order
  GDRabs aux;                    // uv orientation bits / normal flip bits
};
struct GeoDecJob {
  const uint8_t *file; uint32_t file_len; int32_t status;
  int32_t method, traversal;     // encoder_method: 1 edgebreaker (traversal 2 valence / 0 standard), 0 sequential (traversal = connectivity method: 0 compressed / 1 raw)
  uint32_t symbits_off, symbits_n;   // standard traversal: the symbols as an LSB-first bit sequence (byte offset, bits)
  uint32_t seq_idx_off, seq_idx_w;   // sequential, raw indices: byte offset; bytes per index (1, 2, 4) or 0 = varints
  uint32_t ecap;                 // capacity of the per-entry / per-table-vertex arrays (GD_E_WS_OVERFLOW past it: the frame is decoded again with worst-case sizes)
  uint8_t *ws_base; uint64_t ws_zero;   // zero-initialised head of this frame's workspace (k_gdec_clear)
  int32_t nev, nf, nad, nsym, nsplit, nts, ndec, nv, n_interior_start;
  uint32_t ts_off, ts_bits_off;  // topology split events: varint pairs, then packed source-edge bits
  GDRabs rb_start, rb_seam[GD_MAXAD];
  GDRans rs[GD_NRS];
  GDAtt att[GD_MAXDEC];
  int32_t *sp_src, *sp_spl; uint8_t *sp_edge;
  int32_t *opp, *c2v, *lm, *val, *stack, *tsac;
  uint8_t *edge_seam[GD_MAXAD]; int32_t *t_c2v[GD_MAXAD], *t_lm[GD_MAXAD]; int32_t t_nv[GD_MAXAD]; uint8_t *vseam; int32_t *t_cnt; uint8_t *seam_bits;
  uint8_t *vopen[1 + GD_MAXAD];
  uint8_t *aux_bits;             // decoded orientation / flip bits
  int32_t *nbr;                  // parallelogram neighbour entries, 3 per entry and decoder
  uint8_t *uvgeo;                // GDUvGeo per tex-coord entry
  // outputs (device): position / uv / normal values and per-corner entry indices
  float *o_val[3]; uint32_t *o_idx[3]; uint32_t o_n[3]; int32_t o_dec[3];
};

// ---- byte reader (one lane) ----
struct GRd { const uint8_t *b; uint32_t n, o; int err; };
__device__ __forceinline__ uint32_t gr_u8(GRd &r) { if (r.o + 1 > r.n) { r.err = 1; return 0; } return r.b[r.o++]; }
__device__ __forceinline__ int32_t gr_i32(GRd &r) { if (r.o + 4 > r.n) { r.err = 1; return 0; } uint32_t v = 0; for (int k = 0; k < 4; k++) v |= (uint32_t)r.b[r.o + k] << (8 * k); r.o += 4; return (int32_t)v; }
__device__ __forceinline__ float gr_f32(GRd &r) { const int32_t v = gr_i32(r); float f; memcpy(&f, &v, 4); return f; }
__device__ __forceinline__ uint32_t gr_varint(GRd &r) {
  uint64_t v = 0; int s = 0;
  for (;;) { if (r.o >= r.n || s > 63) { r.err = 1; return 0; } const uint32_t c = r.b[r.o++]; v |= (uint64_t)(c & 0x7f) << s; s += 7; if (c < 0x80) break; }
  return (uint32_t)v;
}
// rabs section: prob byte, varint length, payload
__device__ inline void gr_rabs(GRd &r, GDRabs &B) { B.present = 1; B.p0 = gr_u8(r); B.pay_len = gr_varint(r); B.pay_off = r.o; if (r.o + B.pay_len > r.n) r.err = 1; else r.o += B.pay_len; }
// rANS symbol section (RAW scheme): scheme, bit length, alphabet size, probability table, varint length, payload
__device__ inline void gr_rans(GRd &r, GDRans &S, uint32_t nvals) {
  S.present = 1; S.nvals = nvals; S.scheme = gr_u8(r); S.bl = gr_u8(r);
  if (S.scheme != 1) { r.err = 2; return; }
  int pb = (3 * (int)S.bl) / 2; if (pb < 12) pb = 12; if (pb > 20) pb = 20; S.prec_bits = (uint32_t)pb;
  S.ns = gr_varint(r); S.tab_off = r.o;
  if (S.ns > S.max_ns || S.prec_bits > S.max_prec_bits) { r.err = 1; return; }
  for (uint32_t i = 0; i < S.ns && !r.err;) { const uint32_t pd = gr_u8(r), tok = pd & 3; if (tok == 3) i += (pd >> 2) + 1; else { for (uint32_t k = 0; k < tok; k++) (void)gr_u8(r); i++; } }
  S.pay_len = gr_varint(r); S.pay_off = r.o; if (r.o + S.pay_len > r.n) r.err = 1; else r.o += S.pay_len;
}

// ---- sequential connectivity (encoder_method 0; what stock `draco_encoder -cl 0` writes): header, index section, ONE attributes
// decoder with every attribute coded per point in point order (DIFFERENCE predictor or none), transform data of all attributes
// at the end.  Restated from the published bitstream description; no reference fixture uses it. ----
__device__ inline void gd_index_sequential(GeoDecJob &J, GRd &r) {
  const uint32_t n = r.n;
  const int nf = (int)gr_varint(r), np = (int)gr_varint(r), cm = (int)gr_u8(r);
  if (r.err || nf <= 0 || nf != J.nf || np <= 0 || np != J.nev || (cm != 0 && cm != 1)) { J.status = -6; return; }
  if ((uint32_t)np > J.ecap) { J.status = GD_E_WS_OVERFLOW; return; }
  J.traversal = cm; J.nad = 0; J.nsym = 0; J.nsplit = 0; J.nts = 0; J.nv = np;
  for (int i = 0; i < GD_NRS; i++) { J.rs[i].present = 0; J.rs[i].nvals = 0; }
  if (cm == 1) {
    J.seq_idx_off = r.o;
    if (np < 256) J.seq_idx_w = 1; else if (np < (1 << 16)) J.seq_idx_w = 2; else if (np < (1 << 21)) J.seq_idx_w = 0; else J.seq_idx_w = 4;
    if (J.seq_idx_w) { const uint64_t bytes = 3ull * (uint64_t)nf * J.seq_idx_w; if (r.o + bytes > n) { J.status = -8; return; } r.o += (uint32_t)bytes; }
    else { for (int i = 0; i < 3 * nf && !r.err; i++) (void)gr_varint(r); }
  } else gr_rans(r, J.rs[GD_NRS - 1], 3u * (uint32_t)nf);       // index differences as symbols: the last attribute slot's tables (large alphabet)
  if (r.err) { J.status = r.err == 2 ? -25 : -8; return; }
  if (gr_u8(r) != 1) { J.status = -20; return; }
  const int natt = (int)gr_varint(r);
  if (r.err || natt < 1 || natt > GD_MAXDEC || (cm == 0 && natt > GD_MAXDEC - 1)) { J.status = -20; return; }
  J.ndec = natt;
  for (int d = 0; d < natt; d++) { GDAtt &A = J.att[d]; A.att_data_id = -1; A.dec_type = 0; A.table = 0; A.att_type = (int)gr_u8(r); A.data_type = (int)gr_u8(r); A.ncomp = (int)gr_u8(r); (void)gr_u8(r); A.unique_id = (int)gr_varint(r); }
  for (int d = 0; d < natt; d++) J.att[d].seq_type = (int)gr_u8(r);
  if (r.err) { J.status = -22; return; }
  for (int d = 0; d < natt; d++) {
    GDAtt &A = J.att[d];
    A.pred_method = (int8_t)gr_u8(r); A.transform = 0;
    if (A.pred_method != -2) A.transform = (int8_t)gr_u8(r);
    if (gr_u8(r) != 1) { J.status = -24; return; }               // raw (uncompressed) values are not read here
    A.nc = A.seq_type == 3 ? 2 : A.ncomp;
    if (A.nc < 1 || A.nc > 4 || A.seq_type < 1 || A.seq_type > 3) { J.status = -24; return; }
    gr_rans(r, J.rs[6 + d], (uint32_t)np * (uint32_t)A.nc);
    A.aux.present = 0; A.n_orient = 0;
    if (A.pred_method == -2) { }
    else if (A.pred_method == 0 && A.transform == 1) { A.lo = gr_i32(r); A.hi = gr_i32(r); }
    else if (A.pred_method == 0 && A.transform == 3 && A.nc == 2) { A.maxq = gr_i32(r); A.cen = gr_i32(r); }
    else { J.status = -31; return; }
    if (r.err) { J.status = r.err == 2 ? -25 : -32; return; }
  }
  for (int d = 0; d < natt; d++) {
    GDAtt &A = J.att[d];
    if (A.seq_type == 2) { for (int k = 0; k < A.ncomp && k < 4; k++) A.minv[k] = gr_f32(r); A.range = gr_f32(r); A.qbits = (int)gr_u8(r); }
    else if (A.seq_type == 3) A.qbits = (int)gr_u8(r);
  }
  if (r.err) { J.status = -32; return; }
  if (r.o != n) { J.status = -33; return; }
}

// ---- K1: index every section of the file ----
__global__ void __launch_bounds__(64) k_gdec_index(GeoDecJob *jobs) {
  GeoDecJob &J = jobs[blockIdx.x];
  if (threadIdx.x != 0 || J.status != 0) return;
  const uint8_t *b = J.file; const uint32_t n = J.file_len;
  if (n < 11 || b[0] != 'D' || b[1] != 'R' || b[2] != 'A' || b[3] != 'C' || b[4] != 'O') { J.status = -1; return; }
  if (b[7] != 1 || b[8] > 1) { J.status = -2; return; }
  if (b[5] != 2 || b[6] != 2) { J.status = -3; return; }
  if ((b[9] | (b[10] << 8)) != 0) { J.status = -4; return; }
  GRd r; r.b = b; r.n = n; r.o = 11; r.err = 0;
  J.method = b[8];
  if (J.method == 0) { gd_index_sequential(J, r); return; }
  J.traversal = (int)gr_u8(r);
  if (J.traversal != 2 && J.traversal != 0) { J.status = -5; return; }
  const int nev = (int)gr_varint(r), nf = (int)gr_varint(r), nad = (int)gr_u8(r), nsym = (int)gr_varint(r), nsplit = (int)gr_varint(r), nts = (int)gr_varint(r);
  // every count comes from an untrusted varint: the slab is carved for nev + nf + 8 vertices (gdec_carve), so a vertex-split
  // count above nf (one split needs one S symbol, one symbol per face) would let k_gdec_conn write past it
  if (r.err || nf <= 0 || nf != J.nf || nad < 0 || nad > GD_MAXAD || nsym < 0 || nsym > nf || nts < 0 || nts > nf || nsplit < 0 || nsplit > nf || nev < 0 || nev != J.nev) { J.status = -6; return; }
  J.nad = nad; J.nsym = nsym; J.nsplit = nsplit; J.nts = nts;
  if ((uint32_t)nev + (uint32_t)nsplit + 3u > J.ecap) { J.status = GD_E_WS_OVERFLOW; return; }
  { int last = 0; for (int i = 0; i < nts; i++) { const int d = (int)gr_varint(r), src = d + last, d2 = (int)gr_varint(r); J.sp_src[i] = src; J.sp_spl[i] = src - d2; last = src; }
    if (r.o + (uint32_t)(nts + 7) / 8 > n) r.err = 1;
    else { for (int i = 0; i < nts; i++) J.sp_edge[i] = (b[r.o + (i >> 3)] >> (i & 7)) & 1; if (nts > 0) r.o += (uint32_t)(nts + 7) / 8; } }
  if (J.traversal == 0) {                              // standard traversal: size-prefixed bit sequence of the symbols, then start faces and seams
    const uint32_t nb = gr_varint(r); if (r.err || r.o + nb > n || nb > (1u << 28)) { J.status = -8; return; }
    J.symbits_off = r.o; J.symbits_n = 8u * nb; r.o += nb;
  }
  gr_rabs(r, J.rb_start);
  for (int i = 0; i < nad; i++) gr_rabs(r, J.rb_seam[i]);
  for (int i = 0; i < 6; i++) { J.rs[i].present = 0; J.rs[i].nvals = 0; }
  for (int i = 0; i < 6 && J.traversal == 2; i++) { const uint32_t cn = gr_varint(r); if (cn > (uint32_t)nf) r.err = 1; J.rs[i].nvals = cn; if (cn > 0 && !r.err) gr_rans(r, J.rs[i], cn); }
  if (r.err) { J.status = -8; return; }
  // attribute decoder headers (A.4)
  const int ndec = (int)gr_u8(r); if (r.err || ndec < 1 || ndec > GD_MAXDEC) { J.status = -20; return; }
  J.ndec = ndec;
  for (int d = 0; d < ndec; d++) { GDAtt &A = J.att[d]; A.att_data_id = (int8_t)gr_u8(r); A.dec_type = (int)gr_u8(r); if (gr_u8(r) != 0) { J.status = -21; return; } }
  for (int d = 0; d < ndec; d++) {
    GDAtt &A = J.att[d];
    if (gr_varint(r) != 1) { J.status = -22; return; }
    A.att_type = (int)gr_u8(r); A.data_type = (int)gr_u8(r); A.ncomp = (int)gr_u8(r); (void)gr_u8(r); A.unique_id = (int)gr_varint(r);
    A.seq_type = (int)gr_u8(r);
  }
  if (r.err) { J.status = -22; return; }
  // attribute sections: prediction scheme, symbols, scheme data, transform data
  for (int d = 0; d < ndec; d++) {
    GDAtt &A = J.att[d];
    if (A.dec_type == 1) { if (A.att_data_id < 0 || A.att_data_id >= nad || A.att_data_id > 1) { J.status = -23; return; } A.table = 1 + A.att_data_id; } else A.table = 0;
    A.pred_method = (int8_t)gr_u8(r); A.transform = (int8_t)gr_u8(r);
    if (gr_u8(r) != 1) { J.status = -24; return; }
    A.nc = A.seq_type == 3 ? 2 : A.ncomp;
    if (A.nc < 1 || A.nc > 4) { J.status = -24; return; }
    gr_rans(r, J.rs[6 + d], 0);                      // value count = entries * nc, known after the traversal
    A.aux.present = 0; A.n_orient = 0;
    if ((A.pred_method == 1 || A.pred_method == 0) && A.transform == 1) { A.lo = gr_i32(r); A.hi = gr_i32(r); }
    else if (A.pred_method == 5 && A.transform == 1 && A.nc == 2) { A.n_orient = gr_i32(r); if (A.n_orient < 0) { J.status = -26; return; } gr_rabs(r, A.aux); A.lo = gr_i32(r); A.hi = gr_i32(r); }
    else if (A.pred_method == 6 && A.transform == 3 && A.nc == 2) { A.maxq = gr_i32(r); A.cen = gr_i32(r); gr_rabs(r, A.aux); }
    else { J.status = -31; return; }
    if (A.seq_type == 2) { for (int k = 0; k < A.ncomp && k < 4; k++) A.minv[k] = gr_f32(r); A.range = gr_f32(r); A.qbits = (int)gr_u8(r); }
    else if (A.seq_type == 3) A.qbits = (int)gr_u8(r);
    if (r.err) { J.status = r.err == 2 ? -25 : -32; return; }
  }
  if (r.o != n) { J.status = -33; return; }          // every fixture and every file of this codec is consumed to the byte
}

// arrays that start out "invalid" (-1): opposite corners, corner->vertex maps, split map
__global__ void __launch_bounds__(UVOL_BLOCK) k_gdec_init(GeoDecJob *jobs) {
  GeoDecJob &J = jobs[blockIdx.y];
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x, nc = 3u * (uint32_t)J.nf;
  if (i < nc) { J.opp[i] = GEO_INV; J.c2v[i] = GEO_INV; for (int k = 0; k < GD_MAXAD; k++) J.t_c2v[k][i] = GEO_INV; }
  if (i < (uint32_t)J.nf + 2) J.tsac[i] = GEO_INV;
}

// ---- K2: rANS symbol decode, one wave per stream ----
__device__ __forceinline__ int gd_ans_init(const uint8_t *buf, uint32_t n, uint32_t &off, uint32_t &st, uint32_t L, bool allow3) {
  if (n == 0) return -1;
  const int x = buf[n - 1] >> 6; off = n;
  if (x == 0) { st = buf[n - 1] & 0x3f; off -= 1; }
  else if (x == 1) { if (n < 2) return -1; st = ((uint32_t)buf[n - 2] | (uint32_t)buf[n - 1] << 8) & 0x3fff; off -= 2; }
  else if (x == 2) { if (n < 3) return -1; st = ((uint32_t)buf[n - 3] | (uint32_t)buf[n - 2] << 8 | (uint32_t)buf[n - 1] << 16) & 0x3fffff; off -= 3; }
  else { if (!allow3 || n < 4) return -1; st = ((uint32_t)buf[n - 4] | (uint32_t)buf[n - 3] << 8 | (uint32_t)buf[n - 2] << 16 | (uint32_t)buf[n - 1] << 24) & 0x3fffffff; off -= 4; }
  st += L; return 0;
}
// Grid = (frames, streams), frame index fastest: consecutive workgroups land on the four SIMDs of a CU in turn, and with
// (streams, frames) and four streams per frame the one long stream of every frame went to the same SIMD of every CU - a
// quarter of the chip's issue slots for all the serial work of the launch (k_gdec_pred: 0.75 us per entry instead of 0.15)
__global__ void __launch_bounds__(64) k_gdec_rans(GeoDecJob *jobs, int first, int count, int mode) {
  GeoDecJob &J = jobs[blockIdx.x];
  const int si = first + (int)blockIdx.y;
  if ((int)blockIdx.y >= count) return;
  GDRans &S = J.rs[si];
  const uint32_t lane = threadIdx.x;
  // the job status can be changed by the sibling workgroups of this frame (other streams) while this one runs: sample it
  // ONCE per workgroup so that all lanes take the same path to the barriers below
  __shared__ int s_err, s_go;
  // mode 0: every present stream; 1 / 3: the streams whose value count was predicted after the index / after the seam tables (early passes
  // on the second stream); 2: the ones the early passes did not cover or whose count the traversal corrected
  if (lane == 0) s_go = (J.status == 0 && S.present && S.nvals != 0 && (mode == 0 || (mode == 2 ? S.redo != 0 : S.early == (mode == 1 ? 1u : 2u)))) ? 1 : 0;
  __syncthreads();
  if (!s_go) return;
  const uint32_t ns = S.ns, prec = 1u << S.prec_bits, L = prec * 4;
  if (lane == 0) {
    s_err = 0;
    GRd r; r.b = J.file; r.n = J.file_len; r.o = S.tab_off; r.err = 0;
    uint32_t i = 0;
    while (i < ns && !r.err) {
      const uint32_t pd = gr_u8(r), tok = pd & 3;
      if (tok == 3) { uint32_t run = (pd >> 2) + 1; if (i + run > ns) { r.err = 1; break; } while (run--) S.probs[i++] = 0; }
      else { uint32_t p = pd >> 2; for (uint32_t k = 0; k < tok; k++) p |= gr_u8(r) << (8 * (k + 1) - 2); S.probs[i++] = p; }
    }
    uint64_t c = 0;
    for (i = 0; i < ns && !r.err; i++) { S.cum[i] = (uint32_t)c; c += S.probs[i]; if (c > prec) r.err = 1; }
    if (r.err || c != prec) s_err = 1;
  }
  __syncthreads();
  if (s_err) { if (lane == 0) J.status = -5; return; }
  // small tables (the six valence-context streams: 12-bit precision, a handful of symbols) are decoded out of LDS: the chain of a symbol
  // is slot -> symbol -> {frequency, cumulative}, two dependent look-ups that cost an L2 round trip each from global memory
  __shared__ uint16_t l_lut[4096]; __shared__ uint32_t l_probs[256], l_cum[256];
  const bool small = prec <= 4096 && ns <= 256;
  for (uint32_t s = lane; s < ns; s += 64) {
    const uint32_t c = S.cum[s], p = S.probs[s];
    if (small) { l_probs[s] = p; l_cum[s] = c; for (uint32_t j = 0; j < p; j++) l_lut[c + j] = (uint16_t)s; }
    else for (uint32_t j = 0; j < p; j++) S.lut[c + j] = s;
  }
  __threadfence_block();
  __syncthreads();
  if (lane != 0) return;
  const uint8_t *buf = J.file + S.pay_off; uint32_t off, st;
  if (gd_ans_init(buf, S.pay_len, off, st, L, true)) { J.status = -6; return; }
  const uint32_t mask = prec - 1, pb = S.prec_bits, nvals = S.nvals;
  uint32_t *out = S.out;
  // the next payload byte is fetched one step ahead of the renormalisation that consumes it (a byte load behind the state update was one
  // more round trip in the chain of most symbols)
  uint32_t nb = off > 0 ? buf[off - 1] : 0u;
  if (small) {
    for (uint32_t k = 0; k < nvals; k++) {
      while (st < L && off > 0) { off--; st = st * 256 + nb; nb = off > 0 ? buf[off - 1] : 0u; }
      const uint32_t quo = st >> pb, rem = st & mask, s = l_lut[rem];
      st = quo * l_probs[s] + rem - l_cum[s];
      out[k] = s;
    }
  } else {
    const uint32_t *lut = S.lut, *probs = S.probs, *cum = S.cum;
    for (uint32_t k = 0; k < nvals; k++) {
      while (st < L && off > 0) { off--; st = st * 256 + nb; nb = off > 0 ? buf[off - 1] : 0u; }
      const uint32_t quo = st >> pb, rem = st & mask, s = lut[rem];
      st = quo * probs[s] + rem - cum[s];
      out[k] = s;
    }
  }
}

// ---- rabs bit decoder (one lane) ----
struct GDBit { const uint8_t *buf; uint32_t off, st, p0; };
__device__ __forceinline__ int gd_rabs_open(GDBit &R, const GeoDecJob &J, const GDRabs &B) {
  R.buf = J.file + B.pay_off; R.p0 = B.p0;
  if (B.pay_len == 0) { R.st = 4096; R.off = 0; return 0; }
  return gd_ans_init(R.buf, B.pay_len, R.off, R.st, 4096, false);
}
__device__ __forceinline__ int gd_rabs_bit(GDBit &R) {
  const uint32_t p = 256u - R.p0;
  if (R.st < 4096 && R.off > 0) { R.off--; R.st = R.st * 256 + R.buf[R.off]; }
  const uint32_t quot = R.st >> 8, rem = R.st & 255u, xn = quot * p;
  if (rem < p) { R.st = xn + rem; return 1; }
  R.st = R.st - xn - p; return 0;
}

// ---- K3: connectivity, one lane per frame ----
// (STD = the standard traversal's bit-coded symbols; a template so that the valence loop carries neither the test nor the bit reader's registers)
template <bool STD>
__global__ void __launch_bounds__(64) k_gdec_conn(GeoDecJob *jobs) {
  GeoDecJob &J = jobs[blockIdx.x];
  if (threadIdx.x != 0 || J.status != 0 || J.method == 0 || (J.traversal == 0) != STD) return;
  const int nf = J.nf, nsym = J.nsym, nts = J.nts, maxv = J.nev + J.nsplit + 3;
  const uint8_t *sbits = J.file + J.symbits_off; uint32_t sbit = 0; const uint32_t sbit_n = J.symbits_n;
  UVOL_G(int32_t) opp = UVOL_TO_G(int32_t, J.opp); UVOL_G(int32_t) c2v = UVOL_TO_G(int32_t, J.c2v); UVOL_G(int32_t) lm = UVOL_TO_G(int32_t, J.lm);
  UVOL_G(int32_t) val = UVOL_TO_G(int32_t, J.val); UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.stack); UVOL_G(int32_t) tsac = UVOL_TO_G(int32_t, J.tsac);
  UVOL_G(const uint32_t) ctxs[6]; for (int i = 0; i < 6; i++) ctxs[i] = UVOL_TO_G(const uint32_t, J.rs[i].out);
  int cnt[6]; for (int i = 0; i < 6; i++) cnt[i] = (int)J.rs[i].nvals;
  // the next two symbols of every context are kept in registers (the streams are consumed from their ends): the symbol a step needs is
  // chosen by a valence it has only just computed, and a load issued then would be one more round trip in a chain of five
  uint32_t q0[6], q1[6];
#pragma unroll
  for (int i = 0; i < 6; i++) { q0[i] = (!STD && cnt[i] > 0) ? ctxs[i][cnt[i] - 1] : 0u; q1[i] = (!STD && cnt[i] > 1) ? ctxs[i][cnt[i] - 2] : 0u; }
  GDBit SF; if (gd_rabs_open(SF, J, J.rb_start)) { J.status = -7; return; }
  int rc = 0, nv = 0, sp = 0, nfaces = 0, active_ctx = -1, splits_left = nts, n_int = 0;
  int top = GEO_INV;                 // mirror of stack[sp - 1]: the machine reads its own last write most of the time
  int fv0 = 0, fv1 = 0, fv2 = 0;     // vertices of the new face's corners 0, 1, 2 (known without re-reading c2v)
  int lm_fv1 = GEO_INV;              // lm[fv1], fetched together with the valences at the end of a step: a C symbol starts from it, and loading it
                                     // there was the first of three dependent round trips of the step (nothing writes lm[] in between)
  const int SYM2TOPO[5] = { 0, 1, 3, 5, 7 };
#define GD_SETOPP(a, bb) do { opp[a] = (bb); opp[bb] = (a); } while (0)
#define GD_ADDV() (nv < maxv ? (lm[nv] = GEO_INV, nv++) : (rc = -9, 0))
#define GD_BADV(v) ((unsigned)(v) >= (unsigned)nv)          /* vertex id not (yet) allocated: corrupt stream */
#define GD_BADC(c) ((unsigned)(c) >= (unsigned)(3 * nf))
  for (int sid = 0; sid < nsym && !rc; sid++) {
    const int face = nfaces++; int check = 0, sym;
    if (STD) {                                          // 1 bit: C; else two more bits: S 1, L 3, R 5, E 7
      if (sbit + 1 > sbit_n) { rc = -10; break; }
      sym = (sbits[sbit >> 3] >> (sbit & 7)) & 1; sbit++;
      if (sym) { if (sbit + 2 > sbit_n) { rc = -10; break; } for (int k = 0; k < 2; k++, sbit++) sym |= ((sbits[sbit >> 3] >> (sbit & 7)) & 1) << (1 + k); }
    }
    else if (active_ctx != -1) {
      uint32_t s = 0; bool under = false;
#pragma unroll
      for (int i = 0; i < 6; i++) if (active_ctx == i) {          // (unrolled: the per-context registers must not become a scratch array)
        if (--cnt[i] < 0) under = true;
        else { s = q0[i]; q0[i] = q1[i]; q1[i] = cnt[i] > 1 ? ctxs[i][cnt[i] - 2] : 0u; }
      }
      if (under || s > 4) { rc = -10; break; }
      sym = SYM2TOPO[s];
    }
    else sym = 7;
    const int corner = 3 * face;
    if (sym == 0) {
      if (sp == 0) { rc = -11; break; }
      // `top` is always corner 0 of the face added by the previous symbol, whose vertices are still in fv0..fv2: no c2v reads for it
      const int ca = top; if (GD_BADC(ca)) { rc = -11; break; } const int vx = fv1; if (GD_BADV(vx)) { rc = -11; break; }
      const int lmx = lm_fv1; if (GD_BADC(lmx)) { rc = -11; break; }
      const int cb = g_nxt(lmx);
      if (ca == cb || opp[ca] != GEO_INV || opp[cb] != GEO_INV) { rc = -11; break; }
      GD_SETOPP(ca, corner + 1); GD_SETOPP(cb, corner + 2);
      const int vap = fv2, vbn = c2v[g_nxt(cb)]; if (GD_BADV(vap) || GD_BADV(vbn)) { rc = -11; break; }
      c2v[corner] = vx; c2v[corner + 1] = vbn; c2v[corner + 2] = vap; lm[vap] = corner + 2;
      fv0 = vx; fv1 = vbn; fv2 = vap;
      stack[sp - 1] = corner; top = corner;
    } else if (sym == 5 || sym == 3) {
      if (sp == 0) { rc = -12; break; }
      const int ca = top; if (GD_BADC(ca) || opp[ca] != GEO_INV) { rc = -12; break; }
      int oc, cl, cr;
      if (sym == 5) { oc = corner + 2; cl = corner + 1; cr = corner; } else { oc = corner + 1; cl = corner; cr = corner + 2; }
      GD_SETOPP(oc, ca); const int nvx = GD_ADDV(); if (rc) break; c2v[oc] = nvx; lm[nvx] = oc;
      const int vr = fv2, vl = fv1; if (GD_BADV(vr) || GD_BADV(vl)) { rc = -12; break; } c2v[cr] = vr; lm[vr] = cr;
      c2v[cl] = vl;
      if (sym == 5) { fv0 = vr; fv1 = vl; fv2 = nvx; } else { fv0 = vl; fv1 = nvx; fv2 = vr; }
      stack[sp - 1] = corner; top = corner; check = 1;
    } else if (sym == 1) {
      if (sp == 0) { rc = -13; break; }
      const int cb = top; --sp;
      if (tsac[sid] != GEO_INV) { if (sp >= nf + 4) { rc = -13; break; } stack[sp++] = tsac[sid]; }
      if (sp == 0) { rc = -13; break; }
      const int ca = stack[sp - 1];
      if (GD_BADC(ca) || GD_BADC(cb) || ca == cb || opp[ca] != GEO_INV || opp[cb] != GEO_INV) { rc = -13; break; }
      GD_SETOPP(ca, corner + 2); GD_SETOPP(cb, corner + 1);
      const int vp = c2v[g_prv(ca)], vq = c2v[g_nxt(ca)], vbp = c2v[g_prv(cb)]; int cn = g_nxt(cb); const int vn = c2v[cn];
      if (GD_BADV(vp) || GD_BADV(vq) || GD_BADV(vbp) || GD_BADV(vn)) { rc = -13; break; }
      c2v[corner] = vp; c2v[corner + 1] = vq;
      c2v[corner + 2] = vbp; lm[vbp] = corner + 2;
      val[vp] += val[vn]; lm[vp] = lm[vn];
      const int first = cn; int guard = 0;
      while (cn != GEO_INV) { c2v[cn] = vp; const int o2 = opp[g_nxt(cn)]; cn = o2 < 0 ? GEO_INV : g_nxt(o2); if (cn == first || ++guard > 3 * nf) { rc = -13; break; } }
      if (rc) break;
      lm[vn] = GEO_INV;
      fv0 = c2v[corner]; fv1 = c2v[corner + 1]; fv2 = c2v[corner + 2];      // the merge loop above may have re-mapped them
      stack[sp - 1] = corner; top = corner;
    } else {
      const int v0 = GD_ADDV(), v1 = GD_ADDV(), v2 = GD_ADDV(); if (rc) break;
      c2v[corner] = v0; c2v[corner + 1] = v1; c2v[corner + 2] = v2; lm[v0] = corner; lm[v1] = corner + 1; lm[v2] = corner + 2;
      if (sp >= nf + 4) { rc = -13; break; }
      fv0 = v0; fv1 = v1; fv2 = v2;
      stack[sp++] = corner; top = corner; check = 1;
    }
    { // the active corner is corner 0 of the face just added: its vertices are fv0 (corner), fv1 (next), fv2 (prev)
      if (GD_BADV(fv0) || GD_BADV(fv1) || GD_BADV(fv2)) { rc = -18; break; }
      // valence increments of the face's three vertices (C / S: 0 1 1, R: 1 1 2, L: 1 2 1, E: 2 2 2), applied in corner order; the three old
      // values are fetched together and the new valence of fv1 - the next symbol's context - is taken from registers: reading it back
      // after the stores was one more round trip through L2 in the chain (aliases among the three only occur in degenerate streams)
      const int i0 = (sym == 0 || sym == 1) ? 0 : (sym == 7 ? 2 : 1), i1 = sym == 3 || sym == 7 ? 2 : 1, i2 = sym == 5 || sym == 7 ? 2 : 1;
      const int o0 = val[fv0], o1 = val[fv1], o2 = val[fv2];
      lm_fv1 = lm[fv1];
      const int t0 = o0 + i0, t1 = (fv1 == fv0 ? t0 : o1) + i1, t2 = (fv2 == fv1 ? t1 : (fv2 == fv0 ? t0 : o2)) + i2;
      if (i0) val[fv0] = t0;
      val[fv1] = t1; val[fv2] = t2;
      int av = fv2 == fv1 ? t2 : t1; av = av < 2 ? 2 : (av > 7 ? 7 : av); active_ctx = av - 2; }
    if (check) {
      const int esid = nsym - sid - 1;
      while (splits_left > 0 && J.sp_src[splits_left - 1] == esid) {
        splits_left--;
        if (GD_BADC(top)) { rc = -14; break; }
        const int nac = J.sp_edge[splits_left] == 1 ? g_nxt(top) : g_prv(top);
        const int dsid = nsym - J.sp_spl[splits_left] - 1;
        if (dsid < 0 || dsid > nsym) { rc = -14; break; }
        tsac[dsid] = nac;
      }
    }
  }
  while (!rc && sp > 0) {
    const int corner = stack[--sp];
    if (GD_BADC(corner)) { rc = -15; break; }
    if (gd_rabs_bit(SF)) {
      const int vn = c2v[g_nxt(corner)]; if (GD_BADV(vn) || GD_BADC(lm[vn])) { rc = -15; break; }
      const int cb = g_nxt(lm[vn]), vx = c2v[g_nxt(cb)]; if (GD_BADV(vx) || GD_BADC(lm[vx])) { rc = -15; break; }
      const int cc = g_nxt(lm[vx]), vp = c2v[g_nxt(cc)]; if (GD_BADV(vp)) { rc = -15; break; }
      if (nfaces >= nf || opp[corner] != GEO_INV || opp[cb] != GEO_INV || opp[cc] != GEO_INV) { rc = -15; break; }
      const int face = nfaces++, nc = 3 * face;
      GD_SETOPP(nc, corner); GD_SETOPP(nc + 1, cb); GD_SETOPP(nc + 2, cc);
      c2v[nc] = vx; c2v[nc + 1] = vp; c2v[nc + 2] = vn;
      n_int++;
    }
  }
  if (!rc && nfaces != nf) rc = -16;
  for (int i = 0; i < 6 && !rc; i++) if (cnt[i] != 0) rc = -17;
  J.nv = nv; J.n_interior_start = n_int;
  if (rc) J.status = rc;
#undef GD_SETOPP
#undef GD_ADDV
#undef GD_BADV
#undef GD_BADC
}

// every corner must carry an allocated vertex and a symmetric opposite before the parallel stages index with them
__global__ void __launch_bounds__(UVOL_BLOCK) k_gdec_validate(GeoDecJob *jobs, int what) {
  GeoDecJob &J = jobs[blockIdx.y];
  const int c = (int)(blockIdx.x * UVOL_BLOCK + threadIdx.x), nc = 3 * J.nf;
  if (J.status != 0 || c >= nc || J.method == 0) return;
  if (what == 0) {
    const int v = J.c2v[c], o = J.opp[c];
    bool bad = (unsigned)v >= (unsigned)J.nv || o < GEO_INV || o >= nc || (o >= 0 && J.opp[o] != c);
    if (!bad && J.lm[v] == GEO_INV) bad = true;
    if (bad) J.status = -19;
  } else {
    for (int i = 0; i < J.nad; i++) if ((unsigned)J.t_c2v[i][c] >= (unsigned)J.t_nv[i]) J.status = -19;
  }
}

// ---- K4: seam bits (A.5).  One wave per frame: (a) count the edges that carry a bit (opposite face has the larger
// index) with ballots, (b) lanes 0..nad-1 each decode their attribute's rabs stream into a byte array — the only serial
// part —, (c) ballot-ranked assignment of the bits to both corners of each edge, 64 corners at a time. ----
__global__ void __launch_bounds__(64) k_gdec_seams(GeoDecJob *jobs) {
  GeoDecJob &J = jobs[blockIdx.x];
  if (J.status != 0 || J.method == 0) return;
  const uint32_t lane = threadIdx.x;
  const int nc = 3 * J.nf, nad = J.nad; const int32_t *opp = J.opp;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  uint32_t n_elig = 0;
  for (int base = 0; base < nc; base += 64) {
    const int c = base + (int)lane; const int oc = c < nc ? opp[c] : GEO_INV;
    n_elig += (uint32_t)__popcll(__ballot(c < nc && oc != GEO_INV && oc / 3 >= c / 3));
  }
  if ((int)lane < nad) {
    GDBit R; uint8_t *bits = J.seam_bits + (size_t)lane * ((size_t)nc + 64);
    if (gd_rabs_open(R, J, J.rb_seam[lane])) J.status = -7;
    else for (uint32_t k = 0; k < n_elig; k++) bits[k] = (uint8_t)gd_rabs_bit(R);
  }
  __threadfence_block();
  __syncthreads();
  if (J.status != 0) return;
  uint32_t rank0 = 0;
  for (int base = 0; base < nc; base += 64) {
    const int c = base + (int)lane; const int oc = c < nc ? opp[c] : GEO_INV;
    const bool el = c < nc && oc != GEO_INV && oc / 3 >= c / 3;
    const unsigned long long m = __ballot(el);
    if (c < nc && oc == GEO_INV) for (int i = 0; i < nad; i++) J.edge_seam[i][c] = 1;
    if (el) { const uint32_t k = rank0 + (uint32_t)__popcll(m & lt); for (int i = 0; i < nad; i++) if (J.seam_bits[(size_t)i * ((size_t)nc + 64) + k]) { J.edge_seam[i][c] = 1; J.edge_seam[i][oc] = 1; } }
    rank0 += (uint32_t)__popcll(m);
  }
}

// ---- K5: attribute corner tables.  The ids of the attribute vertices around base vertex v are consecutive and start at
// the number of attribute vertices of all earlier base vertices: (a) per-vertex count = 1 + interior seam crossings of its
// fan (thread per vertex), (b) exclusive scan over the vertices (one wave per (attribute, frame)), (c) the same fan walk
// again, writing ids (thread per vertex). ----
__device__ __forceinline__ int gd_fan_first(const GTab &T, const uint8_t *vseam, int v, int c) {
  int first = c;
  if (vseam[v]) { int a = gt_swl(T, first), guard = 0; while (a != GEO_INV) { first = a; a = gt_swl(T, a); if (a == c || ++guard > 4096) break; } }
  return first;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_gdec_vseam(GeoDecJob *jobs) {
  GeoDecJob &J = jobs[blockIdx.y];
  const int i = blockIdx.z, c = (int)(blockIdx.x * UVOL_BLOCK + threadIdx.x);
  if (J.status != 0 || i >= J.nad || c >= 3 * J.nf) return;
  if (J.edge_seam[i][c]) { uint8_t *vseam = J.vseam + (size_t)i * ((size_t)J.nev + J.nf + 72); vseam[J.c2v[g_nxt(c)]] = 1; vseam[J.c2v[g_prv(c)]] = 1; }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_gdec_atttab(GeoDecJob *jobs, int pass) {
  GeoDecJob &J = jobs[blockIdx.y];
  const int i = blockIdx.z, v = (int)(blockIdx.x * UVOL_BLOCK + threadIdx.x);
  if (J.status != 0 || i >= J.nad || v >= J.nv || J.method == 0) return;
  const int32_t *opp = J.opp; const uint8_t *es = J.edge_seam[i];
  const uint8_t *vseam = J.vseam + (size_t)i * ((size_t)J.nev + J.nf + 72);
  int32_t *cntp = J.t_cnt + (size_t)i * ((size_t)J.nev + J.nf + 72);
  const int c = J.lm[v];
  if (c == GEO_INV) { if (pass == 0) cntp[v] = 0; return; }
  GTab T; T.opp = opp; T.seam = es;
  const int first = gd_fan_first(T, vseam, v, c);
  int vid = pass ? cntp[v] : 0, tn = vid + 1;
  if (pass) { J.t_c2v[i][first] = vid; J.t_lm[i][vid] = first; }
  int a = (opp[g_prv(first)] == GEO_INV) ? GEO_INV : g_prv(opp[g_prv(first)]);
  int guard = 0;
  while (a != GEO_INV && a != first) {
    if (++guard > 4096) { J.status = -19; break; }             // fans are short; a cycle that misses `first` means corrupt tables
    if (es[g_nxt(a)]) { vid = tn; if (pass) J.t_lm[i][tn] = a; tn++; }
    if (pass) J.t_c2v[i][a] = vid;
    a = (opp[g_prv(a)] == GEO_INV) ? GEO_INV : g_prv(opp[g_prv(a)]);
  }
  if (pass == 0) cntp[v] = tn;            // number of attribute vertices of v
}
__global__ void __launch_bounds__(64) k_gdec_attscan(GeoDecJob *jobs) {
  GeoDecJob &J = jobs[blockIdx.y];
  const int i = blockIdx.x; const uint32_t lane = threadIdx.x;
  if (J.status != 0 || i >= J.nad) return;
  int32_t *cntp = J.t_cnt + (size_t)i * ((size_t)J.nev + J.nf + 72);
  int run = 0;
  for (int base = 0; base < J.nv; base += 64) {
    const int v = base + (int)lane; int x = v < J.nv ? cntp[v] : 0; const int mine = x;
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
    if (v < J.nv) cntp[v] = run + x - mine;
    run += __shfl(x, 63);
  }
  if (lane == 0) { J.t_nv[i] = run; if ((uint32_t)run > J.ecap) J.status = GD_E_WS_OVERFLOW; }
}

// ---- K6: on-boundary flags per table + the scalars the shared traversal kernels read from their GeoJob ----
__global__ void __launch_bounds__(UVOL_BLOCK) k_gdec_open(GeoDecJob *jobs, GeoJob *gj) {
  GeoDecJob &J = jobs[blockIdx.y];
  GeoJob &G = gj[blockIdx.y];
  const int t = blockIdx.z;                                  // 0 base, 1 + i attribute table
  if (blockIdx.x == 0 && threadIdx.x == 0 && t == 0 && J.method == 0) {      // sequential connectivity: nothing for the traversal kernels to do
    G.status = J.status; G.nf = 0; G.nc = 0; G.nad = 0; G.nverts = 0xffffffffu; G.interior_seams[0] = G.interior_seams[1] = 0;
    for (int k = 0; k < 4; k++) G.nverts_t[k] = 0;
  }
  if (J.method == 0) return;
  if (blockIdx.x == 0 && threadIdx.x == 0 && t == 0) {
    G.status = J.status; G.nf = (uint32_t)J.nf; G.nc = 3u * (uint32_t)J.nf; G.nad = J.nad > 2 ? 2 : J.nad; G.nverts = 0xffffffffu;
    G.nverts_t[1] = (uint32_t)J.nv;
    for (int i = 0; i < 2; i++) { G.interior_seams[i] = i < J.nad ? 1 : 0; G.nverts_t[2 + i] = i < J.nad ? (uint32_t)J.t_nv[i] : 0; }
  }
  if (J.status != 0 || (t > 0 && t - 1 >= J.nad) || t > 2) return;
  const int nvt = t == 0 ? J.nv : J.t_nv[t - 1];
  const int v = (int)(blockIdx.x * UVOL_BLOCK + threadIdx.x);
  if (v >= nvt) return;
  GTab T; T.opp = J.opp; T.seam = t == 0 ? nullptr : J.edge_seam[t - 1];
  const int lmc = t == 0 ? J.lm[v] : J.t_lm[t - 1][v];
  J.vopen[t][v] = (lmc < 0 || gt_swl(T, lmc) < 0) ? 1 : 0;
}

// ---- K9: prediction decode, one lane per (decoder, frame).  phase 0: decoders on the base table (position, generic);
//      phase 1: the ones that need decoded positions (tex-coord-portable, geometric normal) ----
__device__ __forceinline__ int32_t gd_sgn(uint32_t s) { return (s & 1) ? -(int32_t)(s >> 1) - 1 : (int32_t)(s >> 1); }
__device__ __forceinline__ int32_t gd_wrap(int32_t pred, int32_t corr, int32_t lo, int32_t hi) {
  const int32_t md = 1 + hi - lo; int32_t v = (pred < lo ? lo : (pred > hi ? hi : pred)) + corr;
  if (v > hi) v -= md; else if (v < lo) v += md;
  return v;
}
__device__ inline void gd_oct_orig(const GOct &t, const int pred[2], const int corr[2], int32_t out[2]) {
  int ps = pred[0] - t.CEN, pt = pred[1] - t.CEN;
  const bool ind = (g_iabs(ps) + g_iabs(pt)) <= t.CEN;
  if (!ind) g_invert_diamond(t, ps, pt);
  const bool bl = (ps == 0 && pt == 0) || (ps < 0 && pt <= 0);
  const int rc = g_rot_count(ps, pt);
  if (!bl) g_rot(ps, pt, rc);
  int os = g_modmax(t, ps + corr[0]), ot = g_modmax(t, pt + corr[1]);
  if (!bl) g_rot(os, ot, (4 - rc) % 4);
  if (!ind) g_invert_diamond(t, os, ot);
  out[0] = os + t.CEN; out[1] = ot + t.CEN;
}
// parallelogram neighbours of every entry (thread per entry): nb[3p..3p+2] = entries (a, next, prev) of the opposite
// corner when all three are decoded before p, else -1.  Depends only on the connectivity, so the serial recurrence below
// needs no pointer chasing.
__global__ void __launch_bounds__(UVOL_BLOCK) k_gdec_pgram(GeoDecJob *jobs, GeoJob *gj) {
  GeoDecJob &J = jobs[blockIdx.y]; const GeoJob &G = gj[blockIdx.y];
  const int d = blockIdx.z;
  if (J.status != 0 || G.status != 0 || d >= J.ndec) return;
  const GDAtt &A = J.att[d];
  if (!(A.pred_method == 1 || A.pred_method == 0)) return;
  const int t = A.table, p = (int)(blockIdx.x * UVOL_BLOCK + threadIdx.x);
  if (p >= (int)G.ne[t]) return;
  int32_t *nb = J.nbr + (size_t)d * ((size_t)3 * J.ecap + 64) + 3 * (size_t)p;
  nb[0] = nb[1] = nb[2] = -1;
  if (p == 0 || A.pred_method != 1) return;
  const int32_t *v2d = G.v2d[t], *xc2v = t == 0 ? J.c2v : J.t_c2v[t - 1];
  GTab X; X.opp = J.opp; X.seam = t == 0 ? nullptr : J.edge_seam[t - 1];
  const int oci = gt_opp(X, G.order[t][p]);
  if (oci == GEO_INV) return;
  const int a = v2d[xc2v[oci]], bn = v2d[xc2v[g_nxt(oci)]], bp = v2d[xc2v[g_prv(oci)]];
  if (a < p && bn < p && bp < p) { nb[0] = a; nb[1] = bn; nb[2] = bp; }
}

// tex-coord-portable prediction (A.8): everything that depends only on the decoded POSITIONS is computed per entry in
// parallel — neighbour entries, |pn|^2, the projection dot product and the integer square root —; the serial recurrence
// keeps the two uv reads, four multiplies and two truncating divisions.  uvg[p] = { nd, pd, pn2, dd, ns }.
struct GDUvGeo { int32_t nd, pd; long long pn2, dd, ns; };
__global__ void __launch_bounds__(UVOL_BLOCK) k_gdec_uvgeo(GeoDecJob *jobs, GeoJob *gj) {
  GeoDecJob &J = jobs[blockIdx.y]; const GeoJob &G = gj[blockIdx.y];
  const int d = blockIdx.z;
  if (J.status != 0 || G.status != 0 || d >= J.ndec) return;
  const GDAtt &A = J.att[d];
  if (A.pred_method != 5) return;
  const int t = A.table, p = (int)(blockIdx.x * UVOL_BLOCK + threadIdx.x);
  if (p >= (int)G.ne[t]) return;
  int pdec = -1; for (int k = 0; k < J.ndec; k++) if (J.att[k].att_type == 0 && J.att[k].att_data_id == -1) pdec = k;
  if (pdec < 0) return;
  const int32_t *P = J.att[pdec].vals, *b_v2d = G.v2d[0], *c2v = J.c2v, *v2d = G.v2d[t], *xc2v = t == 0 ? J.c2v : J.t_c2v[t - 1];
  const int c = G.order[t][p], cn = g_nxt(c), cp = g_prv(c);
  GDUvGeo g; g.nd = v2d[xc2v[cn]]; g.pd = v2d[xc2v[cp]]; g.pn2 = 0; g.dd = 0; g.ns = 0;
  if (g.pd < p && g.nd < p) {
    const int32_t *tip = P + 3 * b_v2d[c2v[c]], *np_ = P + 3 * b_v2d[c2v[cn]], *pp_ = P + 3 * b_v2d[c2v[cp]];
    long long pn[3], pn2 = 0, dd = 0;
    for (int k = 0; k < 3; k++) { pn[k] = (long long)pp_[k] - np_[k]; pn2 += pn[k] * pn[k]; }
    if (pn2 != 0) {
      for (int k = 0; k < 3; k++) dd += pn[k] * ((long long)tip[k] - np_[k]);
      long long cx2 = 0;
      for (int k = 0; k < 3; k++) { const long long xp = np_[k] + (dd * pn[k]) / pn2, e = tip[k] - xp; cx2 += e * e; }
      g.ns = (long long)g_isqrt((uint64_t)cx2 * (uint64_t)pn2);
    }
    g.pn2 = pn2; g.dd = dd;
  }
  reinterpret_cast<GDUvGeo *>(J.uvgeo)[p] = g;
}

// geometric-normal prediction (A.9): independent per entry once the flip bits are known -> thread per entry
__global__ void __launch_bounds__(UVOL_BLOCK) k_gdec_normals(GeoDecJob *jobs, GeoJob *gj) {
  GeoDecJob &J = jobs[blockIdx.y]; const GeoJob &G = gj[blockIdx.y];
  const int d = blockIdx.z;
  if (J.status != 0 || G.status != 0 || d >= J.ndec) return;
  GDAtt &A = J.att[d];
  if (A.pred_method != 6) return;
  const int t = A.table, dd = (int)(blockIdx.x * UVOL_BLOCK + threadIdx.x);
  if (dd >= (int)G.ne[t]) return;
  int pdec = -1; for (int k = 0; k < J.ndec; k++) if (J.att[k].att_type == 0 && J.att[k].att_data_id == -1) pdec = k;
  if (pdec < 0) return;
  const int32_t *P = J.att[pdec].vals, *b_v2d = G.v2d[0], *c2v = J.c2v;
  GTab X; X.opp = J.opp; X.seam = t == 0 ? nullptr : J.edge_seam[t - 1];
  int q = 0; while ((1 << q) - 1 < A.maxq) q++;
  const GOct ot = g_oct(q);
  const uint32_t *syms = J.rs[6 + d].out; const uint8_t *flips = J.aux_bits + (size_t)d * ((size_t)J.ecap + 64);
  const int c0 = G.order[t][dd];
  const int32_t *cenp = P + 3 * b_v2d[c2v[c0]];
  long long N[3] = { 0, 0, 0 };
  int c = c0, guard = 0; bool left = true;
  while (c != GEO_INV && ++guard <= 4096) {
    const int32_t *a = P + 3 * b_v2d[c2v[g_nxt(c)]], *bb = P + 3 * b_v2d[c2v[g_prv(c)]];
    long long dn[3], dp[3];
    for (int k = 0; k < 3; k++) { dn[k] = (long long)a[k] - cenp[k]; dp[k] = (long long)bb[k] - cenp[k]; }
    N[0] += dn[1] * dp[2] - dn[2] * dp[1]; N[1] += dn[2] * dp[0] - dn[0] * dp[2]; N[2] += dn[0] * dp[1] - dn[1] * dp[0];
    if (left) { c = gt_swl(X, c); if (c == c0) break; if (c == GEO_INV) { left = false; c = gt_swr(X, c0); } }
    else c = gt_swr(X, c);
  }
  long long s = g_labs(N[0]) + g_labs(N[1]) + g_labs(N[2]);
  if (s > (1 << 29)) { const long long qd = s / (1 << 29); for (int k = 0; k < 3; k++) N[k] /= qd; }
  int pv[3];
  s = g_labs(N[0]) + g_labs(N[1]) + g_labs(N[2]);
  if (s == 0) { pv[0] = ot.CEN; pv[1] = 0; pv[2] = 0; }
  else { const long long aa = (N[0] * ot.CEN) / s, bb2 = (N[1] * ot.CEN) / s; long long cc = ot.CEN - g_labs(aa) - g_labs(bb2); if (N[2] < 0) cc = -cc; pv[0] = (int)aa; pv[1] = (int)bb2; pv[2] = (int)cc; }
  if (flips[dd]) { pv[0] = -pv[0]; pv[1] = -pv[1]; pv[2] = -pv[2]; }
  int po[2]; g_vec_to_oct(ot, pv, po[0], po[1]);
  const int corr[2] = { (int)syms[2 * dd], (int)syms[2 * dd + 1] };
  gd_oct_orig(ot, po, corr, A.vals + 2 * dd);
}

// Value recurrences of one decoder (A.6 parallelogram + wrap, A.8 tex-coord-portable), one wave per (decoder, frame), 64 entries
// at a time.  An entry's operands are earlier OUTPUTS; read back from global memory right after the store, each costs a round
// trip through L2 (the chain ran at ~1.3 us per entry).  Per chunk of 64 entries:
//   stage (64 lanes): neighbour entries / geometry record / symbols of the lane's entry; operands older than the chunk are
//                     fetched now - from the LDS ring of the last GDP_R outputs, or from global memory when older than that
//                     (written at least four chunks ago) - and only operands INSIDE the chunk stay as references;
//   chain (lane 0):   the 64 entries in order, operands from the staged record or the chunk's LDS tile;
//   store (64 lanes): tile -> out[] (coalesced) and -> ring.
#define GDP_R 256
#define GDP_STAGE 16      // words per staged entry
template <int NC>
__device__ __forceinline__ void gd_pgram_chunks(int ne, UVOL_G(const int32_t) nbr, UVOL_G(const uint32_t) syms, UVOL_G(int32_t) out, int32_t lo, int32_t hi,
                                                int32_t *ring, int32_t *stage, int32_t *tile) {
  // (typed global pointers: with generic ones every load is a flat_load the compiler must drain before the next LDS store,
  //  and the staging of a chunk became a dozen round trips in a row)
  const int lane = (int)threadIdx.x;
  for (int base = 0; base < ne; base += 64) {
    const int e = base + lane;
    if (e < ne) {
      const int a = nbr[3 * (size_t)e], bn = nbr[3 * (size_t)e + 1], bp = nbr[3 * (size_t)e + 2];
      int32_t cr[NC];
#pragma unroll
      for (int k = 0; k < NC; k++) cr[k] = gd_sgn(syms[(size_t)e * NC + k]);
      int q[3] = { -1, -1, -1 };                          // operands: +q0 +q1 -q2
      if (a >= 0) { q[0] = bn; q[1] = bp; q[2] = a; } else if (e > 0) q[0] = e - 1;
      int32_t ps[NC]; uint32_t refs = 0;
#pragma unroll
      for (int k = 0; k < NC; k++) ps[k] = 0;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        if (q[i] < 0) continue;
        if (q[i] >= base) { refs |= (uint32_t)(q[i] - base + 1) << (8 * i); continue; }
        if (q[i] < base - GDP_R) {
#pragma unroll
          for (int k = 0; k < NC; k++) { const int32_t v = out[(size_t)q[i] * NC + k]; ps[k] += i == 2 ? -v : v; }
        } else {
#pragma unroll
          for (int k = 0; k < NC; k++) { const int32_t v = ring[(q[i] & (GDP_R - 1)) * NC + k]; ps[k] += i == 2 ? -v : v; }
        }
      }
      int32_t *st = stage + lane * GDP_STAGE;
#pragma unroll
      for (int k = 0; k < NC; k++) { st[k] = ps[k]; st[4 + k] = cr[k]; }
      st[8] = (int32_t)refs;
    }
    __syncthreads();
    if (lane == 0) {
      // the chain: per entry one round trip to the LDS tile (all three possible references are read at once, unused ones masked)
      // and the wrap; the staged record of the next entry is requested before this one is computed
      const int cnt = ne - base < 64 ? ne - base : 64;
      int32_t ps[NC], cr[NC]; uint32_t refs = (uint32_t)stage[8];
#pragma unroll
      for (int k = 0; k < NC; k++) { ps[k] = stage[k]; cr[k] = stage[4 + k]; }
      for (int j = 0; j < cnt; j++) {
        const int32_t *sn = stage + (j + 1 < cnt ? j + 1 : j) * GDP_STAGE;
        int32_t nps[NC], ncr[NC]; const uint32_t nrefs = (uint32_t)sn[8];
#pragma unroll
        for (int k = 0; k < NC; k++) { nps[k] = sn[k]; ncr[k] = sn[4 + k]; }
        const uint32_t r0 = refs & 255u, r1 = (refs >> 8) & 255u, r2 = refs >> 16;
        const int32_t *t0 = tile + (r0 ? r0 - 1 : 0) * NC, *t1 = tile + (r1 ? r1 - 1 : 0) * NC, *t2 = tile + (r2 ? r2 - 1 : 0) * NC;
        int32_t a0[NC], a1[NC], a2[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) { a0[k] = t0[k]; a1[k] = t1[k]; a2[k] = t2[k]; }
#pragma unroll
        for (int k = 0; k < NC; k++) tile[j * NC + k] = gd_wrap(ps[k] + (r0 ? a0[k] : 0) + (r1 ? a1[k] : 0) - (r2 ? a2[k] : 0), cr[k], lo, hi);
#pragma unroll
        for (int k = 0; k < NC; k++) { ps[k] = nps[k]; cr[k] = ncr[k]; }
        refs = nrefs;
      }
    }
    __syncthreads();
    if (e < ne) {
#pragma unroll
      for (int k = 0; k < NC; k++) { const int32_t v = tile[lane * NC + k]; out[(size_t)e * NC + k] = v; ring[(size_t)(e & (GDP_R - 1)) * NC + k] = v; }
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(64) k_gdec_pred(GeoDecJob *jobs, GeoJob *gj, int phase) {
  GeoDecJob &J = jobs[blockIdx.x];
  const GeoJob &G = gj[blockIdx.x];
  const int d = blockIdx.y, lane = (int)threadIdx.x;                  // (frame index fastest: see k_gdec_rans)
  __shared__ int s_go, s_bad;
  __shared__ __attribute__((aligned(16))) int32_t s_ring[GDP_R * 4], s_stage[64 * GDP_STAGE], s_tile[64 * 4];
  // the job status can be changed by the sibling workgroups of this frame while this one runs: sample it once per workgroup
  if (lane == 0) { s_go = (J.status == 0 && G.status == 0 && d < J.ndec) ? 1 : 0; s_bad = 0; }
  __syncthreads();
  if (!s_go) return;
  GDAtt &A = J.att[d];
  if (A.pred_method == 6) return;                       // normals: k_gdec_flips + k_gdec_normals
  const bool needs_pos = A.pred_method == 5;
  if ((phase == 0) == needs_pos) return;
  const int t = A.table, nc = A.nc, ne = (int)G.ne[t];
  UVOL_G(const uint32_t) syms = UVOL_TO_G(const uint32_t, J.rs[6 + d].out); UVOL_G(int32_t) out = UVOL_TO_G(int32_t, A.vals);
  int pdec = -1; for (int k = 0; k < J.ndec; k++) if (J.att[k].att_type == 0 && J.att[k].att_data_id == -1) pdec = k;
  if (A.pred_method == -2) { for (int i = lane; i < ne * nc; i += 64) out[i] = gd_sgn(syms[i]); }          // no prediction (sequential streams)
  else if (A.pred_method == 0 && A.transform == 3) {                                                // DIFFERENCE through the canonicalised octahedron (sequential normals)
    if (lane != 0) return;
    int q = 0; while ((1 << q) - 1 < A.maxq) q++;
    const GOct ot = g_oct(q);
    if (ot.MAXQ != A.maxq || ot.CEN != A.cen || nc != 2) { J.status = -30; return; }
    int prev[2] = { 0, 0 };
    for (int p = 0; p < ne; p++) { const int corr[2] = { (int)syms[2 * p], (int)syms[2 * p + 1] }; int32_t o2[2]; gd_oct_orig(ot, prev, corr, o2); out[2 * p] = o2[0]; out[2 * p + 1] = o2[1]; prev[0] = o2[0]; prev[1] = o2[1]; }
  }
  else if (A.pred_method == 1 || A.pred_method == 0) {
    const int32_t lo = A.lo, hi = A.hi;
    UVOL_G(const int32_t) nbr = UVOL_TO_G(const int32_t, J.nbr + (size_t)d * ((size_t)3 * J.ecap + 64));
    // component count as a template parameter: the per-component arrays must stay in registers (a run-time bound sends
    // them to scratch memory)
    if (nc == 3) gd_pgram_chunks<3>(ne, nbr, syms, out, lo, hi, s_ring, s_stage, s_tile);
    else if (nc == 1) gd_pgram_chunks<1>(ne, nbr, syms, out, lo, hi, s_ring, s_stage, s_tile);
    else if (nc == 2) gd_pgram_chunks<2>(ne, nbr, syms, out, lo, hi, s_ring, s_stage, s_tile);
    else gd_pgram_chunks<4>(ne, nbr, syms, out, lo, hi, s_ring, s_stage, s_tile);
  } else if (A.pred_method == 5) {
    const int no = A.n_orient; uint8_t *ori = J.aux_bits + (size_t)d * ((size_t)J.ecap + 64);
    if (lane == 0) {
      if (pdec < 0 || (uint32_t)no > J.ecap) { J.status = -26; s_bad = 1; }                // (more orientation bits than entries: corrupt)
      else { GDBit Rb; if (gd_rabs_open(Rb, J, A.aux)) { J.status = -26; s_bad = 1; } else { int last = 1; for (int k = 0; k < no; k++) { if (!gd_rabs_bit(Rb)) last = !last; ori[k] = (uint8_t)last; } } }
    }
    __syncthreads();
    if (s_bad) return;
    const int32_t lo = A.lo, hi = A.hi; int nori = no;
    UVOL_G(const int32_t) uvw = UVOL_TO_G(const int32_t, reinterpret_cast<const int32_t *>(J.uvgeo));           // GDUvGeo records as 8 words
    UVOL_G(const uint8_t) orig = UVOL_TO_G(const uint8_t, ori);
    __shared__ uint8_t s_ori[64]; __shared__ int s_nori;
    // f64 form of the prediction (below): exact while every product stays below 2^53, which quantisation up to 16 bits guarantees
    // (|uv| < 2^17, pn2 / dd / ns < 2^35); anything larger - only a corrupt stream - takes the int64 form
    const bool small_uv = lo > -(1 << 17) && hi < (1 << 17);
    if (lane == 0) s_nori = no;
    // staged entry: [0] flags (1: nd < p, 2: pd < p too, 4: a value to fall back on, 8: f64 form is exact), [1] in-chunk references
    // (nd + 1) | (pd + 1) << 8, [2..3] nuv, [4..5] puv, [6..7] corrections, [8..9] pn2, [10..11] dd, [12..13] ns, [14..15] 1 / pn2 (f64)
    for (int base = 0; base < ne; base += 64) {
      const int e = base + lane;
      __syncthreads();
      { const int k = s_nori - 1 - lane; s_ori[lane] = k >= 0 ? orig[k] : 0; }          // the (at most 64) orientation bits this chunk can consume
      if (e < ne) {
        GDUvGeo g;
        { uint32_t u[8];
#pragma unroll
          for (int k = 0; k < 8; k++) u[k] = (uint32_t)uvw[8 * (size_t)e + k];
          g.nd = (int32_t)u[0]; g.pd = (int32_t)u[1]; g.pn2 = (long long)(((unsigned long long)u[3] << 32) | u[2]); g.dd = (long long)(((unsigned long long)u[5] << 32) | u[4]); g.ns = (long long)(((unsigned long long)u[7] << 32) | u[6]); }
        const int32_t sy0 = gd_sgn(syms[2 * (size_t)e]), sy1 = gd_sgn(syms[2 * (size_t)e + 1]);
        int32_t *st = s_stage + lane * GDP_STAGE;
        const bool hn = (uint32_t)g.nd < (uint32_t)e, hp = hn && (uint32_t)g.pd < (uint32_t)e;      // (negative = an entry the tables never reached: corrupt input)
        uint32_t refs = 0; int32_t nuv[2] = { 0, 0 }, puv[2] = { 0, 0 };
        const int qn = hn ? g.nd : (e > 0 ? e - 1 : -1);       // without a decoded 'next' vertex the previous entry predicts
        if (qn >= 0) {
          if (qn >= base) refs |= (uint32_t)(qn - base + 1);
          else if (qn < base - GDP_R) { nuv[0] = out[2 * (size_t)qn]; nuv[1] = out[2 * (size_t)qn + 1]; }
          else { nuv[0] = s_ring[2 * (qn & (GDP_R - 1))]; nuv[1] = s_ring[2 * (qn & (GDP_R - 1)) + 1]; }
        }
        if (hp) {
          if (g.pd >= base) refs |= (uint32_t)(g.pd - base + 1) << 8;
          else if (g.pd < base - GDP_R) { puv[0] = out[2 * (size_t)g.pd]; puv[1] = out[2 * (size_t)g.pd + 1]; }
          else { puv[0] = s_ring[2 * (g.pd & (GDP_R - 1))]; puv[1] = s_ring[2 * (g.pd & (GDP_R - 1)) + 1]; }
        }
        const bool fast = small_uv && g.pn2 > 0 && g.pn2 < (1ll << 35) && g.dd > -(1ll << 35) && g.dd < (1ll << 35) && g.ns >= 0 && g.ns < (1ll << 35);
        st[0] = (hn ? 1 : 0) | (hp ? 2 : 0) | (qn >= 0 ? 4 : 0) | (fast ? 8 : 0); st[1] = (int32_t)refs; st[2] = nuv[0]; st[3] = nuv[1]; st[4] = puv[0]; st[5] = puv[1];
        st[6] = sy0; st[7] = sy1;
        st[8] = (int32_t)(uint32_t)g.pn2; st[9] = (int32_t)(g.pn2 >> 32); st[10] = (int32_t)(uint32_t)g.dd; st[11] = (int32_t)(g.dd >> 32); st[12] = (int32_t)(uint32_t)g.ns; st[13] = (int32_t)(g.ns >> 32);
        const double rp = fast ? 1.0 / (double)g.pn2 : 0.0;
        *reinterpret_cast<double *>(st + 14) = rp;
      }
      __syncthreads();
      if (lane == 0) {
        const int cnt = ne - base < 64 ? ne - base : 64;
        int used = 0;
        for (int j = 0; j < cnt; j++) {
          const int32_t *st = s_stage + j * GDP_STAGE;
          const uint32_t fl = (uint32_t)st[0], refs = (uint32_t)st[1];
          const uint32_t rn = refs & 255u, rp_ = refs >> 8;
          // both possible references and the next orientation bit in one LDS round trip
          const int32_t tn0 = s_tile[2 * (rn ? rn - 1 : 0)], tn1 = s_tile[2 * (rn ? rn - 1 : 0) + 1], tp0 = s_tile[2 * (rp_ ? rp_ - 1 : 0)], tp1 = s_tile[2 * (rp_ ? rp_ - 1 : 0) + 1];
          const int o_ = s_ori[used & 63];
          const int32_t nuv[2] = { rn ? tn0 : st[2], rn ? tn1 : st[3] }, puv[2] = { rp_ ? tp0 : st[4], rp_ ? tp1 : st[5] };
          const long long pn2 = *reinterpret_cast<const long long *>(st + 8), dd = *reinterpret_cast<const long long *>(st + 10), ns_ = *reinterpret_cast<const long long *>(st + 12);
          const bool both = (fl & 2u) != 0, same = puv[0] == nuv[0] && puv[1] == nuv[1], par = both && !same && pn2 != 0;
          int32_t pred[2] = { (fl & 4u) ? nuv[0] : 0, (fl & 4u) ? nuv[1] : 0 };       // the 'next' vertex's value or the previous entry's (both && same: equal to puv)
          if (par) {
            if (nori <= 0) { J.status = -27; s_bad = 1; break; }
            nori--; used++;
            if (fl & 8u) {
              // (xuv +- cxuv) / pn2, truncated: all terms are integers below 2^53, so the f64 sums are exact; the quotient estimate
              // through the staged reciprocal is within one of the truth and the remainder (one fma, exact) settles it
              const double P = (double)pn2, D = (double)dd, NS = (double)ns_, RP = *reinterpret_cast<const double *>(st + 14);
              const double pu = (double)(puv[0] - nuv[0]), pv = (double)(puv[1] - nuv[1]);
              const double cs = o_ ? NS : -NS;
              const double n0 = fma((double)nuv[0], P, fma(D, pu, pv * cs)), n1 = fma((double)nuv[1], P, fma(D, pv, -(pu * cs)));
#pragma unroll
              for (int k = 0; k < 2; k++) {
                const double nk = k ? n1 : n0;
                double q = trunc(nk * RP); const double r = fma(-q, P, nk);
                const double up = nk >= 0 ? (r >= P ? 1.0 : 0.0) : (r > 0 ? 1.0 : 0.0), dn = nk >= 0 ? (r < 0 ? 1.0 : 0.0) : (r <= -P ? 1.0 : 0.0);
                q += up - dn;
                pred[k] = (int32_t)q;
              }
            } else {
              const long long pnuv[2] = { (long long)puv[0] - nuv[0], (long long)puv[1] - nuv[1] };
              const long long xuv[2] = { nuv[0] * pn2 + dd * pnuv[0], nuv[1] * pn2 + dd * pnuv[1] };
              const long long cxuv[2] = { pnuv[1] * ns_, -pnuv[0] * ns_ };
              if (o_) { pred[0] = (int32_t)((xuv[0] + cxuv[0]) / pn2); pred[1] = (int32_t)((xuv[1] + cxuv[1]) / pn2); }
              else { pred[0] = (int32_t)((xuv[0] - cxuv[0]) / pn2); pred[1] = (int32_t)((xuv[1] - cxuv[1]) / pn2); }
            }
          }
          s_tile[2 * j] = gd_wrap(pred[0], st[6], lo, hi); s_tile[2 * j + 1] = gd_wrap(pred[1], st[7], lo, hi);
        }
        s_nori = nori;
      }
      __syncthreads();
      if (s_bad) return;
      if (e < ne) { const int32_t v0 = s_tile[2 * lane], v1 = s_tile[2 * lane + 1]; out[2 * (size_t)e] = v0; out[2 * (size_t)e + 1] = v1; s_ring[2 * (e & (GDP_R - 1))] = v0; s_ring[2 * (e & (GDP_R - 1)) + 1] = v1; }
    }
    if (lane == 0 && nori != 0) J.status = -28;
  }
}

// sequential connectivity: the point index of every corner -> c2v[] (one wave per frame; varint- and difference-coded indices by lane 0)
__global__ void __launch_bounds__(64) k_gdec_seq_conn(GeoDecJob *jobs) {
  GeoDecJob &J = jobs[blockIdx.x];
  if (J.status != 0 || J.method != 0) return;
  const int nc = 3 * J.nf, np = J.nv; const uint32_t lane = threadIdx.x;
  bool bad = false;
  if (J.traversal == 1 && J.seq_idx_w) {
    const uint8_t *p = J.file + J.seq_idx_off; const uint32_t w = J.seq_idx_w;
    for (int i = (int)lane; i < nc; i += 64) { uint32_t v = 0; for (uint32_t k = 0; k < w; k++) v |= (uint32_t)p[(size_t)i * w + k] << (8 * k); if (v >= (uint32_t)np) bad = true; J.c2v[i] = (int32_t)v; }
  } else if (lane == 0) {
    if (J.traversal == 1) { GRd r; r.b = J.file; r.n = J.file_len; r.o = J.seq_idx_off; r.err = 0; for (int i = 0; i < nc; i++) { const uint32_t v = gr_varint(r); if (v >= (uint32_t)np) bad = true; J.c2v[i] = (int32_t)v; } if (r.err) bad = true; }
    else { const uint32_t *sy = J.rs[GD_NRS - 1].out; int32_t last = 0; for (int i = 0; i < nc; i++) { int32_t d = (int32_t)(sy[i] >> 1); if (sy[i] & 1) d = -d; last += d; if ((uint32_t)last >= (uint32_t)np) bad = true; J.c2v[i] = last; } }
  }
  if (bad) J.status = -19;
}

// flip bits of the normal decoders: the only sequential part of the geometric-normal scheme (one lane per decoder)
__global__ void __launch_bounds__(64) k_gdec_flips(GeoDecJob *jobs, GeoJob *gj) {
  GeoDecJob &J = jobs[blockIdx.x]; const GeoJob &G = gj[blockIdx.x];
  const int d = blockIdx.y;
  if (threadIdx.x != 0 || J.status != 0 || G.status != 0 || d >= J.ndec) return;
  GDAtt &A = J.att[d];
  if (A.pred_method != 6) return;
  int q = 0; while ((1 << q) - 1 < A.maxq) q++;
  const GOct ot = g_oct(q);
  if (ot.MAXQ != A.maxq || ot.CEN != A.cen) { J.status = -30; return; }
  GDBit Fb; if (gd_rabs_open(Fb, J, A.aux)) { J.status = -29; return; }
  uint8_t *flips = J.aux_bits + (size_t)d * ((size_t)J.ecap + 64);
  const int ne = (int)G.ne[A.table];
  for (int k = 0; k < ne; k++) flips[k] = (uint8_t)gd_rabs_bit(Fb);
}

// Attribute symbol counts = entries of the decoder's table x components.  The traversal counts the entries, but a valid file's counts are
// known before it: the header's vertex count for the base table, the attribute vertices the seam tables produced for the others.  The
// symbol streams (serial rANS chains, 220 ms per 1920 frames) are therefore decoded on a second stream WHILE the traversals (200 ms) run;
// k_gdec_counts then compares, and a stream whose count the traversal corrected (a header that lies) is decoded again.
// phase 0 (after the index): the streams of the base table, whose count is in the header - they run beside the connectivity decoder already;
// phase 1 (after the seam tables): the streams of the attribute tables
__global__ void __launch_bounds__(64) k_gdec_counts_early(GeoDecJob *jobs, int phase) {
  GeoDecJob &J = jobs[blockIdx.x];
  if (threadIdx.x != 0 || J.status != 0) return;
  // sequential connectivity: one entry per point, known from the header; the compressed index differences (connectivity method 0) sit in the
  // LAST slot, which no attribute decoder uses then (ndec <= GD_MAXDEC - 1) - it has to be marked too or no pass decodes it
  if (J.method == 0) { if (phase == 0) { for (int d = 0; d < J.ndec; d++) J.rs[6 + d].early = 1; J.rs[GD_NRS - 1].early = 1; } return; }
  for (int d = 0; d < J.ndec; d++) {
    const int t = J.att[d].table;
    if ((t == 0) != (phase == 0)) continue;
    const uint32_t ne = t == 0 ? (uint32_t)J.nev : (t - 1 < J.nad ? (uint32_t)J.t_nv[t - 1] : 0u);
    GDRans &S = J.rs[6 + d];
    if (ne != 0 && ne <= J.ecap) { S.nvals = ne * (uint32_t)J.att[d].nc; S.early = 1u + (uint32_t)phase; }
  }
}
__global__ void __launch_bounds__(64) k_gdec_counts(GeoDecJob *jobs, GeoJob *gj) {
  GeoDecJob &J = jobs[blockIdx.x]; GeoJob &G = gj[blockIdx.x];
  if (threadIdx.x != 0) return;
  if (G.status != 0 && J.status == 0) J.status = G.status;
  if (J.status != 0) return;
  if (J.method == 0) { G.ne[0] = (uint32_t)J.nv; for (int d = 0; d < J.ndec; d++) J.rs[6 + d].redo = J.rs[6 + d].early ? 0u : 1u; J.rs[GD_NRS - 1].redo = J.rs[GD_NRS - 1].early ? 0u : 1u; return; }      // sequential: one entry per point, the value counts were known at once
  for (int d = 0; d < J.ndec; d++) {
    GDRans &S = J.rs[6 + d];
    const uint32_t nv = G.ne[J.att[d].table] * (uint32_t)J.att[d].nc;
    S.redo = (S.early && S.nvals == nv) ? 0u : 1u;
    S.nvals = nv;
  }
}

// ---- K10: outputs.  blockIdx.z = 0 position, 1 tex-coord, 2 normal ----
__global__ void __launch_bounds__(UVOL_BLOCK) k_gdec_finish(GeoDecJob *jobs, GeoJob *gj) {
  GeoDecJob &J = jobs[blockIdx.y]; const GeoJob &G = gj[blockIdx.y];
  if (J.status != 0) return;
  const int which = blockIdx.z;
  int d = -1;
  for (int k = 0; k < J.ndec; k++) { const int at = J.att[k].att_type; if ((which == 0 && at == 0) || (which == 1 && at == 3) || (which == 2 && at == 1)) { d = k; break; } }
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (d < 0) { if (i == 0) { J.o_n[which] = 0; J.o_dec[which] = -1; } return; }
  const GDAtt &A = J.att[d];
  const int t = A.table; const uint32_t ne = G.ne[t];
  if (i == 0) { J.o_n[which] = ne; J.o_dec[which] = d; }
  if (i < 3u * (uint32_t)J.nf && J.o_idx[which]) { const int32_t *xc2v = t == 0 ? J.c2v : J.t_c2v[t - 1]; J.o_idx[which][i] = J.method == 0 ? (uint32_t)J.c2v[i] : (uint32_t)G.v2d[t][xc2v[i]]; }
  if (i >= ne || !J.o_val[which]) return;
  float *o = J.o_val[which];
  if (A.seq_type == 2) {
    const float delta = A.range / (float)((1u << A.qbits) - 1);
    for (int k = 0; k < A.ncomp; k++) o[(size_t)i * A.ncomp + k] = A.minv[k] + (float)A.vals[(size_t)i * A.ncomp + k] * delta;
  } else if (A.seq_type == 3) {
    const GOct ot = g_oct(A.qbits);
    float y = (float)A.vals[2 * (size_t)i] * (2.0f / (float)ot.MAXV) - 1.0f, z = (float)A.vals[2 * (size_t)i + 1] * (2.0f / (float)ot.MAXV) - 1.0f;
    float x = 1.0f - fabsf(y) - fabsf(z); const float xo = x < 0 ? -x : 0;
    y += y < 0 ? xo : -xo; z += z < 0 ? xo : -xo;
    const float nn = sqrtf(x * x + y * y + z * z);
    if (nn > 1e-6f) { x /= nn; y /= nn; z /= nn; } else { x = y = z = 0; }
    o[3 * (size_t)i] = x; o[3 * (size_t)i + 1] = y; o[3 * (size_t)i + 2] = z;
  } else for (int k = 0; k < A.ncomp; k++) o[(size_t)i * A.ncomp + k] = (float)A.vals[(size_t)i * A.ncomp + k];
}

// ================================================================================================
// host side
// ================================================================================================
struct GDPlan { std::vector<uint64_t> key; std::vector<size_t> offs; size_t total = 0, zero = 0; };      // workspace placement of the last frame dimensions seen (gdec_carve)
struct GeoDecState { uvol_devbuf files, slab, jobs, gjobs, outs; std::vector<GeoDecJob> hjobs; std::vector<GeoJob> hg; GDPlan plan;
                     hipStream_t aux = nullptr; hipEvent_t ev_tabs[2] = { nullptr, nullptr }, ev_sym = nullptr; };      // aux: the attribute symbol streams, beside the traversals
int geodec_create(uvol_ctx *ctx) { ctx->geodec = new GeoDecState(); return UVOL_OK; }
void geodec_destroy(uvol_ctx *ctx) {
  GeoDecState *t = ctx->geodec; if (!t) return;
  for (uvol_devbuf *b : { &t->files, &t->slab, &t->jobs, &t->gjobs, &t->outs }) if (b->p) (void)hipFree(b->p);
  if (t->aux) { (void)hipStreamSynchronize(t->aux); (void)hipStreamDestroy(t->aux); }
  for (int k = 0; k < 2; k++) if (t->ev_tabs[k]) (void)hipEventDestroy(t->ev_tabs[k]);
  if (t->ev_sym) (void)hipEventDestroy(t->ev_sym);
  delete t; ctx->geodec = nullptr;
}

static bool gdec_header(const uint8_t *b, size_t n, uint32_t *nev, uint32_t *nf) {
  if (!b || n < 16 || memcmp(b, "DRACO", 5) || b[8] > 1) return false;
  size_t o = b[8] == 0 ? 11 : 12; uint32_t v[2];                          // sequential: faces, points; edgebreaker: traversal byte, vertices, faces
  for (int k = 0; k < 2; k++) { uint64_t r = 0; int s = 0; for (;;) { if (o >= n || s > 35) return false; const uint8_t c = b[o++]; r |= (uint64_t)(c & 0x7f) << s; s += 7; if (c < 0x80) break; } v[k] = (uint32_t)r; }
  if (b[8] == 0) { const uint32_t t = v[0]; v[0] = v[1]; v[1] = t; if (v[0] == 0 || v[0] > 3 * v[1]) return false; }
  *nev = v[0]; *nf = v[1];
  return v[1] > 0 && v[1] <= (1u << 26) && v[0] <= 3 * v[1] + 8 && (uint64_t)v[1] <= 16ull * n;      // a face costs > 1/16 byte
}
extern "C" int uvol_drc_info(const uint8_t *drc, size_t len, uint32_t *n_faces, uint32_t *max_values) {
  uint32_t nev = 0, nf = 0;
  if (!gdec_header(drc, len, &nev, &nf)) return UVOL_E_INVALID;
  if (n_faces) *n_faces = nf;
  if (max_values) *max_values = 3 * nf;
  return UVOL_OK;
}

// geom_encode.hip: k_pack_faces + k_traverse + k_v2d over tables 1..3 of a GeoJob array (the encoder's own sequencing kernels)
int geo_run_traversals(uvol_ctx *ctx, GeoJob *gj, int n, uint32_t max_nfi, uint32_t max_vals);
bool geo_records8(uint32_t max_nfi);

#define GLAUNCH(k, grid, block, shmem, ...)                                                      \
  do {                                                                                           \
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch %s\n", #k); fflush(stderr); }              \
    hipLaunchKernelGGL(k, grid, block, shmem, ctx->stream, __VA_ARGS__);                         \
    if (uvol_debug()) { hipError_t e_ = hipStreamSynchronize(ctx->stream); if (e_ != hipSuccess) { fprintf(stderr, "[uvol] %s FAILED: %s\n", #k, hipGetErrorString(e_)); fflush(stderr); } } \
  } while (0)

// zero the head of every frame's workspace (the arrays that must start out zero: valences, seam flags, visited bitmaps)
__global__ void __launch_bounds__(UVOL_BLOCK) k_gdec_clear(GeoDecJob *jobs) {
  GeoDecJob &J = jobs[blockIdx.y];
  uint4 *p = reinterpret_cast<uint4 *>(J.ws_base);
  const size_t n16 = (size_t)(J.ws_zero / 16);
  for (size_t i = (size_t)blockIdx.x * UVOL_BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * UVOL_BLOCK) p[i] = make_uint4(0, 0, 0, 0);
}

// Workspace of one frame (base == nullptr: size only); also wires the GeoJob view the shared traversal kernels read (table 1 =
// base corner table, 2 / 3 = attribute tables 0 / 1).  Every array carries the first and last stage that touches it and arrays
// with disjoint lifetimes share addresses (uvol_ws.hpp); the arrays that must start out zero form the pinned head.  Per-entry
// arrays (traversal order, predictor scratch, symbols, values) are sized for `ecap` entries - faces + faces / 2 in the compact
// layout, a frame that needs more (every corner its own entry) fails with GD_E_WS_OVERFLOW on the device and is decoded again
// with the worst case (3 x faces).  265 -> ~90 MB per 200 k-face frame.
enum { DS_INDEX = 0, DS_CTX, DS_CONN, DS_TABLES, DS_TRAV, DS_SYM, DS_PRED, DS_FIN, DS_COUNT };
static size_t gdec_carve(GeoDecJob &J, GeoJob &G, uint8_t *base, bool r8, bool full, GDPlan &P) {
  const size_t nf = (size_t)J.nf, nc = 3 * nf, maxv = (size_t)J.nev + nf + 8;          // nsplit <= nf
  const size_t E = full ? nc + 3 : std::min(nc + 3, nf + nf / 2 + 4096);
  J.ecap = (uint32_t)E;
  std::vector<UvolWsItem> items; std::vector<void **> slots;
#define DCARVE(field, bytes, first, last) do { items.push_back(UvolWsItem{ (size_t)(bytes), (first), (last), 0 }); slots.push_back((void **)&(field)); } while (0)
  DCARVE(J.sp_src, 4 * (nf + 1), DS_INDEX, DS_CONN); DCARVE(J.sp_spl, 4 * (nf + 1), DS_INDEX, DS_CONN); DCARVE(J.sp_edge, nf + 8, DS_INDEX, DS_CONN);
  DCARVE(J.opp, 4 * nc, DS_INDEX, DS_FIN); DCARVE(J.c2v, 4 * nc, DS_INDEX, DS_FIN);
  DCARVE(J.lm, 4 * maxv, DS_CONN, DS_TABLES); DCARVE(J.val, 4 * maxv, UVOL_WS_PINNED, UVOL_WS_PINNED);
  DCARVE(J.stack, 4 * (nf + 8), DS_CONN, DS_CONN); DCARVE(J.tsac, 4 * (nf + 2), DS_INDEX, DS_CONN);
  for (int k = 0; k < GD_MAXAD; k++) { DCARVE(J.edge_seam[k], nc, UVOL_WS_PINNED, UVOL_WS_PINNED); DCARVE(J.t_c2v[k], 4 * nc, DS_INDEX, DS_FIN); DCARVE(J.t_lm[k], 4 * E, DS_TABLES, DS_TABLES); }
  DCARVE(J.vseam, GD_MAXAD * (maxv + 64), UVOL_WS_PINNED, UVOL_WS_PINNED); DCARVE(J.t_cnt, 4 * GD_MAXAD * (maxv + 64), DS_TABLES, DS_TABLES); DCARVE(J.seam_bits, GD_MAXAD * (nc + 64), DS_TABLES, DS_TABLES);
  for (int k = 0; k < 1 + GD_MAXAD; k++) DCARVE(J.vopen[k], std::max(E, maxv) + 64, DS_TABLES, DS_TRAV);
  DCARVE(J.aux_bits, GD_MAXDEC * (E + 64), DS_PRED, DS_PRED); DCARVE(J.nbr, 4 * GD_MAXDEC * (3 * E + 64), DS_PRED, DS_PRED); DCARVE(J.uvgeo, 32 * (E + 8), DS_PRED, DS_PRED);
  for (int k = 0; k < 6; k++) { GDRans &S = J.rs[k]; S.max_ns = GD_CTX_NS; S.max_prec_bits = 12; DCARVE(S.probs, 4 * GD_CTX_NS, DS_CTX, DS_CONN); DCARVE(S.cum, 4 * GD_CTX_NS, DS_CTX, DS_CONN); DCARVE(S.lut, 4 * (1u << 12), DS_CTX, DS_CONN); DCARVE(S.out, 4 * (nf + 1), DS_CTX, DS_CONN); }
  for (int k = 0; k < GD_MAXDEC; k++) {
    GDRans &S = J.rs[6 + k]; S.max_ns = GD_MAX_NS; S.max_prec_bits = 20;
    // (written from the traversal stage on: the early pass of the symbol decoder runs beside the traversals)
    DCARVE(S.probs, 4 * (size_t)GD_MAX_NS, DS_CTX, DS_SYM); DCARVE(S.cum, 4 * (size_t)GD_MAX_NS, DS_CTX, DS_SYM); DCARVE(S.lut, 4 * (size_t)(1u << 20), DS_CTX, DS_SYM);
    // (the last slot also holds the index differences of a frame with compressed sequential connectivity: one per corner)
    DCARVE(S.out, 4 * (std::max(4 * E, k == GD_MAXDEC - 1 ? nc : (size_t)0) + 4), DS_CTX, DS_PRED); DCARVE(J.att[k].vals, 4 * (4 * E + 4), DS_PRED, DS_FIN);
  }
  for (int k = 1; k < 4; k++) DCARVE(G.rec[k], (r8 ? 32 : 64) * (nf + 1), DS_TRAV, DS_TRAV);     // 8- or 16-byte corner records, decided per batch (geo_records8)
  for (int k = 0; k < 3; k++) { DCARVE(G.order[k], 4 * (E + 3), DS_TRAV, DS_FIN); DCARVE(G.v2d[k], 4 * (std::max(E, maxv) + 3), DS_TRAV, DS_FIN); DCARVE(G.t_stack[k], 4 * (nf + 2), DS_TRAV, DS_TRAV); DCARVE(G.t_vvis[k], std::max(E, maxv) / 8 + 64, UVOL_WS_PINNED, UVOL_WS_PINNED); DCARVE(G.t_fvis[k], nf / 8 + 64, UVOL_WS_PINNED, UVOL_WS_PINNED); }
#undef DCARVE
  const std::vector<uint64_t> key = { (uint64_t)nf, (uint64_t)J.nev, (uint64_t)r8 | ((uint64_t)full << 1), (uint64_t)items.size() };
  if (key != P.key) { P.total = uvol_ws_place(items, &P.zero, DS_COUNT, "geometry decode"); P.offs.resize(items.size()); for (size_t i = 0; i < items.size(); i++) P.offs[i] = items[i].off; P.key = key; }
  for (size_t i = 0; i < slots.size(); i++) *slots[i] = base ? (void *)(base + P.offs[i]) : nullptr;
  J.ws_base = base; J.ws_zero = P.zero;
  G.status = 0; G.nf = (uint32_t)nf; G.nc = (uint32_t)nc; G.nad = 2; G.nverts = 0xffffffffu; G.ecap = (uint32_t)E;
  G.nopp = J.opp; G.bvert = J.c2v; G.avert[0] = J.t_c2v[0]; G.avert[1] = J.t_c2v[1]; G.seam[0] = J.edge_seam[0]; G.seam[1] = J.edge_seam[1];
  G.vopen_d[1] = J.vopen[0]; G.vopen_d[2] = J.vopen[1]; G.vopen_d[3] = J.vopen[2];
  return P.total;
}

static int geo_decode_batch_impl(uvol_ctx *ctx, const uint8_t *const *files, const size_t *lens, int n, uvol_decoded_mesh *out, int *status, bool full, bool out_dev);
int geo_decode_batch(uvol_ctx *ctx, const uint8_t *const *files, const size_t *lens, int n, uvol_decoded_mesh *out, int *status, bool outputs_on_device) {
  return geo_decode_batch_impl(ctx, files, lens, n, out, status, false, outputs_on_device);
}
static int geo_decode_batch_impl(uvol_ctx *ctx, const uint8_t *const *files, const size_t *lens, int n, uvol_decoded_mesh *out, int *status, bool full, bool out_dev) {
  GeoDecState *T = ctx->geodec;
  if (n <= 0) return UVOL_OK;
  T->hjobs.assign((size_t)n, GeoDecJob{}); T->hg.assign((size_t)n, GeoJob{});
  std::vector<size_t> foff((size_t)n), woff((size_t)n), ooff((size_t)n);
  size_t ftot = 0, wtot = 0, otot = 0; uint32_t max_nf = 0, max_nev = 0;
  auto a256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  bool r8 = true;                                                        // record format of the batch: every frame's face count must allow 8-byte records
  { uint32_t mf = 0; for (int i = 0; i < n; i++) { uint32_t nev = 0, nf = 0; if (gdec_header(files[i], lens[i], &nev, &nf)) mf = std::max(mf, nf); } r8 = geo_records8(mf); }
  for (int i = 0; i < n; i++) {
    uint32_t nev = 0, nf = 0;
    if (!gdec_header(files[i], lens[i], &nev, &nf)) { ctx->set_error("frame %d: not a Draco 2.2 edgebreaker mesh", i); return UVOL_E_INVALID; }
    if (out[i].cap_faces < nf || out[i].cap_values < 3 * (size_t)nf) { ctx->set_error("frame %d: output capacity too small (%u faces)", i, nf); return UVOL_E_NOSPACE; }
    GeoDecJob &J = T->hjobs[i]; J.nf = (int32_t)nf; J.nev = (int32_t)nev; J.file_len = (uint32_t)lens[i];
    max_nf = std::max(max_nf, nf); max_nev = std::max(max_nev, nev);
    foff[i] = ftot; ftot += a256(lens[i] + 16);
    GeoJob gtmp{}; const size_t w = gdec_carve(J, gtmp, nullptr, r8, full, T->plan);
    woff[i] = wtot; wtot += a256(w);
    { const size_t nc = 3 * (size_t)nf; ooff[i] = otot; otot += 3 * (a256(4 * 3 * nc) + a256(4 * nc)); }
  }
  int rc;
  if ((rc = uvol_ensure(ctx, T->files, ftot + 64))) return rc;
  if ((rc = uvol_ensure(ctx, T->slab, wtot))) return rc;
  if ((rc = uvol_ensure(ctx, T->jobs, sizeof(GeoDecJob) * (size_t)n))) return rc;
  if ((rc = uvol_ensure(ctx, T->gjobs, sizeof(GeoJob) * (size_t)n))) return rc;
  if ((rc = uvol_ensure(ctx, T->outs, otot))) return rc;
  std::vector<UvolUpItem> ups; ups.reserve((size_t)n);
  for (int i = 0; i < n; i++) {
    GeoDecJob &J = T->hjobs[i]; GeoJob &G = T->hg[i];
    uint8_t *fd = (uint8_t *)T->files.p + foff[i];
    ups.push_back(UvolUpItem{ foff[i], files[i], lens[i] });      // one staged upload for the whole call below (1920 pageable copies cost ~0.1 s)
    J.file = fd; J.status = 0;
    (void)gdec_carve(J, G, (uint8_t *)T->slab.p + woff[i], r8, full, T->plan);
    const size_t nc = 3 * (size_t)J.nf;
    uint8_t *ob = (uint8_t *)T->outs.p + ooff[i]; size_t oo = 0;
    for (int k = 0; k < 3; k++) { J.o_val[k] = (float *)(ob + oo); oo += a256(4 * 3 * nc); J.o_idx[k] = (uint32_t *)(ob + oo); oo += a256(4 * nc); }
  }
  { const int rcu = uvol_upload_staged(ctx, (uint8_t *)T->files.p, ups); if (rcu != UVOL_OK) return rcu; }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(T->jobs.p, T->hjobs.data(), sizeof(GeoDecJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(T->gjobs.p, T->hg.data(), sizeof(GeoJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  GeoDecJob *dj = (GeoDecJob *)T->jobs.p; GeoJob *gj = (GeoJob *)T->gjobs.p;
  const unsigned N = (unsigned)n, bc = uvol_blocks((size_t)3 * max_nf);
  GLAUNCH(k_gdec_clear, dim3(64, N), dim3(UVOL_BLOCK), 0, dj);
  { uvol_ctx::Scope sc(ctx, "geodec.k1_index", 0); GLAUNCH(k_gdec_init, dim3(bc, N), dim3(UVOL_BLOCK), 0, dj); GLAUNCH(k_gdec_index, dim3(N), dim3(64), 0, dj); }
  // the attribute symbol streams on the second stream: the base table's beside the connectivity decoder, the others beside the traversals
  if (!T->aux) {
    UVOL_HIP_CHECK(ctx, hipStreamCreateWithFlags(&T->aux, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) UVOL_HIP_CHECK(ctx, hipEventCreateWithFlags(&T->ev_tabs[k], hipEventDisableTiming));      // (one per phase: a wait must not see a later record of its event)
    UVOL_HIP_CHECK(ctx, hipEventCreateWithFlags(&T->ev_sym, hipEventDisableTiming));
  }
  // any early return from here to the end-of-call synchronisation leaves no kernel running on the second stream (they write J.status and
  // S.out; a retry or the next call lays T->jobs / T->slab out again on ctx->stream with no ordering against it)
  struct AuxGuard { hipStream_t s; bool armed; ~AuxGuard() { if (armed) (void)hipStreamSynchronize(s); } } aux_guard{ T->aux, true };
  auto early_pass = [&](int phase) -> int {
    UVOL_HIP_CHECK(ctx, hipEventRecord(T->ev_tabs[phase], ctx->stream));
    UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(T->aux, T->ev_tabs[phase], 0));
    hipStream_t main = ctx->stream; ctx->stream = T->aux;
    GLAUNCH(k_gdec_counts_early, dim3(N), dim3(64), 0, dj, phase);
    GLAUNCH(k_gdec_rans, dim3(N, GD_MAXDEC), dim3(64), 0, dj, 6, GD_MAXDEC, phase == 0 ? 1 : 3);
    ctx->stream = main;
    return UVOL_OK; };
  if ((rc = early_pass(0))) return rc;
  { uvol_ctx::Scope sc(ctx, "geodec.k2_ctx_symbols", 0); GLAUNCH(k_gdec_rans, dim3(N, 6), dim3(64), 0, dj, 0, 6, 0); }
  { uvol_ctx::Scope sc(ctx, "geodec.k3_connectivity", 0); GLAUNCH(k_gdec_conn<false>, dim3(N), dim3(64), 0, dj); GLAUNCH(k_gdec_conn<true>, dim3(N), dim3(64), 0, dj); GLAUNCH(k_gdec_validate, dim3(bc, N), dim3(UVOL_BLOCK), 0, dj, 0); }
  { uvol_ctx::Scope sc(ctx, "geodec.k4_seams_tables", 0);
    GLAUNCH(k_gdec_seams, dim3(N), dim3(64), 0, dj);
    GLAUNCH(k_gdec_vseam, dim3(bc, N, GD_MAXAD), dim3(UVOL_BLOCK), 0, dj);
    GLAUNCH(k_gdec_atttab, dim3(bc, N, GD_MAXAD), dim3(UVOL_BLOCK), 0, dj, 0);
    GLAUNCH(k_gdec_attscan, dim3(GD_MAXAD, N), dim3(64), 0, dj);
    GLAUNCH(k_gdec_atttab, dim3(bc, N, GD_MAXAD), dim3(UVOL_BLOCK), 0, dj, 1);
    GLAUNCH(k_gdec_validate, dim3(bc, N), dim3(UVOL_BLOCK), 0, dj, 1);
    GLAUNCH(k_gdec_open, dim3(bc, N, 3), dim3(UVOL_BLOCK), 0, dj, gj); }
  if ((rc = early_pass(1))) return rc;
  UVOL_HIP_CHECK(ctx, hipEventRecord(T->ev_sym, T->aux));
  { uvol_ctx::Scope sc(ctx, "geodec.k5_traverse", 0);
    if ((rc = geo_run_traversals(ctx, gj, n, max_nf, max_nev + max_nev / 4 + 64))) return rc;
    UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, T->ev_sym, 0));
    GLAUNCH(k_gdec_counts, dim3(N), dim3(64), 0, dj, gj); }
  { uvol_ctx::Scope sc(ctx, "geodec.k6_attr_symbols", 0); GLAUNCH(k_gdec_rans, dim3(N, GD_MAXDEC), dim3(64), 0, dj, 6, GD_MAXDEC, 2);
    GLAUNCH(k_gdec_seq_conn, dim3(N), dim3(64), 0, dj); }                  // frames with sequential connectivity: their index section
  { uvol_ctx::Scope sc(ctx, "geodec.k7_predict", 0);
    GLAUNCH(k_gdec_pgram, dim3(bc, N, GD_MAXDEC), dim3(UVOL_BLOCK), 0, dj, gj);
    GLAUNCH(k_gdec_flips, dim3(N, GD_MAXDEC), dim3(64), 0, dj, gj);
    GLAUNCH(k_gdec_pred, dim3(N, GD_MAXDEC), dim3(64), 0, dj, gj, 0);
    GLAUNCH(k_gdec_normals, dim3(bc, N, GD_MAXDEC), dim3(UVOL_BLOCK), 0, dj, gj);
    GLAUNCH(k_gdec_uvgeo, dim3(bc, N, GD_MAXDEC), dim3(UVOL_BLOCK), 0, dj, gj);
    GLAUNCH(k_gdec_pred, dim3(N, GD_MAXDEC), dim3(64), 0, dj, gj, 1); }
  { uvol_ctx::Scope sc(ctx, "geodec.k8_finish", 0); GLAUNCH(k_gdec_finish, dim3(bc, N, 3), dim3(UVOL_BLOCK), 0, dj, gj); }
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(T->hjobs.data(), dj, sizeof(GeoDecJob) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  aux_guard.armed = false;                                 // (the main stream waited for ev_sym: the second stream is idle)
  if (uvol_debug()) { int ne = 0, nr = 0; for (int i = 0; i < n; i++) for (int d = 0; d < T->hjobs[i].ndec; d++) { ne += T->hjobs[i].rs[6 + d].early != 0; nr += T->hjobs[i].rs[6 + d].redo != 0; }
    fprintf(stderr, "[uvol] decode: %d attribute streams decoded beside the traversal, %d (again) after it\n", ne, nr); }
  int worst = UVOL_OK;
  std::vector<int> retry;
  std::vector<UvolDnItem> dns; if (!out_dev) dns.reserve((size_t)n * 6);      // host outputs: ONE staged download for the whole call (uvol_download_staged)
  for (int i = 0; i < n; i++) {
    const GeoDecJob &J = T->hjobs[i]; uvol_decoded_mesh &M = out[i];
    if (!full && J.status == GD_E_WS_OVERFLOW) { retry.push_back(i); if (status) status[i] = UVOL_OK; continue; }      // decoded again below, alone, with worst-case sizes
    const int st = J.status == 0 ? UVOL_OK : UVOL_E_ENCODE;
    if (status) status[i] = st;
    if (st != UVOL_OK) { ctx->set_error("frame %d: corrupt or unsupported .drc (device status %d)", i, J.status); worst = st; continue; }
    M.n_faces = (uint32_t)J.nf;
    float *vals[3] = { M.pos, M.uv, M.nrm }; uint32_t *idx[3] = { M.idx_pos, M.idx_uv, M.idx_nrm }; uint32_t *cnt[3] = { &M.n_pos, &M.n_uv, &M.n_nrm };
    const int comps[3] = { 3, 2, 3 };
    for (int k = 0; k < 3; k++) {
      *cnt[k] = J.o_n[k];
      if (!J.o_n[k]) continue;
      if (out_dev) {                                                       // (uvol_decode_mesh_batch_dev: the arrays stay in HBM)
        if (vals[k]) UVOL_HIP_CHECK(ctx, hipMemcpyAsync(vals[k], J.o_val[k], (size_t)J.o_n[k] * comps[k] * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (idx[k]) UVOL_HIP_CHECK(ctx, hipMemcpyAsync(idx[k], J.o_idx[k], (size_t)J.nf * 3 * 4, hipMemcpyDeviceToDevice, ctx->stream));
      } else {
        if (vals[k]) dns.push_back(UvolDnItem{ J.o_val[k], vals[k], (size_t)J.o_n[k] * comps[k] * 4 });
        if (idx[k]) dns.push_back(UvolDnItem{ J.o_idx[k], idx[k], (size_t)J.nf * 3 * 4 });
      }
    }
  }
  if (!out_dev) { const int rcd = uvol_download_staged(ctx, dns); if (rcd != UVOL_OK) return rcd; }
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->resolve_profile();
  for (int i : retry) {                                   // frames the compact workspace could not hold (more entries per face than usual)
    int st1 = UVOL_OK;
    const int rc1 = geo_decode_batch_impl(ctx, files + i, lens + i, 1, out + i, &st1, true, out_dev);
    if (rc1 != UVOL_OK) return rc1;
    if (status) status[i] = st1;
    if (st1 != UVOL_OK) worst = st1;
  }
  return status ? UVOL_OK : worst;
}
