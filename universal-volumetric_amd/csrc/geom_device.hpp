// geom_device.hpp — device-side data model + small device helpers of the geometry (.drc) pipeline.
//
// Replaces the arithmetic of one `draco_encoder -qp -qt -qn -cl 7` process (scripts/Encoder.py:260)
// with a batch of frames resident in HBM.  Bitstream layout: SURVEY.md Appendix A (Draco 2.2,
// edgebreaker + valence traversal); encoder-side rules: SURVEY.md A.10.
//
// Data layout in HBM: one GeoJob per frame; every per-corner array is a flat SoA int32/uint8
// array of 3*F entries so that consecutive lanes touch consecutive addresses in the parallel
// kernels (dedup, corner table, fans, seams, quantise, predictors, histograms, gather).  The three
// inherently serial walkers (edgebreaker, attribute DFS, rANS/rabs state machines) run one frame /
// stream per workgroup, so a batch of N frames keeps N (or 9N / 5N) CUs busy at once.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#define GEO_NSTREAM 9     // rANS streams: 6 valence contexts, position, uv, normal
#define GEO_NRABS 5       // rabs streams: start faces, seam uv, seam normal, uv orientations, normal flips
#define GEO_MAXPIECES 64

struct RansStream {
  const uint32_t *syms;   // symbols (device)
  uint32_t n;             // number of symbols
  uint32_t *freq;         // [alpha_cap] histogram (zeroed per batch)
  uint32_t *probs;        // [alpha_cap]
  uint32_t *cum;          // [alpha_cap]
  uint32_t *scratch;      // counting-sort scratch of k_rans_tables
  uint4 *tab;             // [alpha_cap] {prob | shift << 24, cum, reciprocal, 0} per symbol (lane-per-stream coder)
  uint32_t alpha_cap;
  uint32_t max_sym;       // atomicMax target
  uint32_t prec_bits;
  uint8_t *head;  uint32_t head_len;      // scheme, bl, varint(ns), table
  uint8_t *pay;   uint32_t pay_cap;       // [8 bytes reserved for varint(len)] payload
  uint32_t pay_off, pay_len;              // piece = pay + pay_off, length pay_len (varint + payload)
};

struct RabsStream {
  const uint8_t *bits; uint32_t n; uint32_t zeros;
  uint8_t *buf; uint32_t cap; uint32_t off, len;   // piece = buf + off
};

struct GeoJob {
  // ---- inputs (device pointers) ----
  const float *pos, *uv, *nrm;
  const uint32_t *ipos, *iuv, *inrm;
  uint32_t n_pos, n_uv, n_nrm, nf_in;
  int32_t has_uv, has_nrm, nad;
  int32_t qp, qt, qn;
  // ---- state ----
  int32_t status;
  uint32_t nf, nc, nverts;
  int32_t nsym, nsplit, nev, nstart, ninit;
  uint32_t interior_seams[2];
  int32_t att_kind[2];          // 0 uv, 1 normal, per attribute-data slot
  uint32_t n_elig;              // seam-bit count
  uint32_t ne[3];               // entries: base, att0, att1 (att uses base when no interior seams)
  uint32_t n_ori, ne_uv, ne_nrm;
  uint32_t pos_min_u[3], pos_max_u[3], uv_min_u[2], uv_max_u[2];   // orderable-float encodings
  int32_t wrap_lo[2], wrap_hi[2];                                  // [0]=pos, [1]=uv
  uint32_t out_len;
  // ---- workspace ----
  uint32_t *dd_tab[3]; uint32_t dd_cap[3]; uint32_t *canon[3]; uint32_t n_dup[3];      // hash-table dedup (worst-case retry): n_dup: phase 0 saw two equal values
  uint4 *dd_part[3]; uint32_t *dd_cnt[3]; uint32_t dd_nb[3], dd_nblk[3];               // partitioned dedup: {index, words} records by hash bin; counts[bin][tile]
  // locality relabelling (k_ms_*): positions get new ids in Morton order of their quantised coordinates, faces are stored in the
  // order of their lowest new vertex id.  Ids and storage order are identities only - the bitstream is the one the input order
  // gives (component starts and non-manifold tie-breaks still follow the ORIGINAL face order through forig / s_of_o).
  int32_t compact;                         // 1: laid out without stored value-id copies and relabelling scratch (clean, coherently stored frames only: geo_submit_impl)
  uint32_t n_degen;                        // faces with two equal position indices (k_coherence; meaningful while no two positions are equal)
  int32_t relabel;                         // 1: on for this frame (2 while undecided: k_coherence / k_relabel_decide)
  uint32_t coh_share, coh_tight, coh_same; // k_coherence: faces sharing a vertex with their predecessor / with a narrow index span / equal to the same face of the previous frame
  uint32_t *ms_key[2];                     // [0] Morton key per position; [1] per input face: lowest new vertex id (~0u: dropped face)
  uint2 *ms_part; uint32_t *ms_cnt;        // {key, index} records by bin; counts[bin][tile]
  uint32_t ms_nb[2], ms_nblk[2], ms_sh[2]; // bins, tiles, bin = key >> sh
  uint32_t *prank;                         // position -> new id
  float *pos_s;                            // positions in new-id order
  uint32_t *fperm, *cidx;                  // sorted slot -> input face; input face -> its index among the kept faces (original order)
  int32_t *forig, *s_of_o;                 // stored face -> original kept-face index, and back
  // sequential connectivity (DRACO_COMPRESSION_LEVEL 0, k_sq_*): points = distinct (position, uv, normal) value triples in order
  // of first appearance over the corners; a 64-bit-keyed hash table finds the first corner of every pair, twice
  int32_t late_join;        // small batches: the auxiliary stream (valence replay) is joined before the entropy stage, not before the record tables (its inputs then keep their own bytes)
  int32_t seq; uint32_t sq_cap, sq_np, sq_idx_bytes;
  unsigned long long *sq_keys; uint32_t *sq_val;
  int32_t *sq_pu, *sq_first, *sq_pid, *sq_cop;       // per corner: first corner with the same (pos, uv) / the same triple; point id; per point: its first corner
  uint8_t *sq_flag; uint8_t *sq_idx;                  // scan flags / per-corner index byte counts; the index section's bytes
  uint32_t *he_part, *he_cnt; uint32_t he_vpb, he_nb, he_nblk;      // partitioned bucket build: {from, to, corner} records by vertex range; counts[bin][tile]; vertices per bin (0: atomic build)
  uint32_t *he_start, *he_cur; unsigned long long *he_ent;   // half-edges bucketed by their from-vertex: [he_start[a], he_cur[a]) holds (to-vertex << 32 | corner)
  uint8_t *keep; uint32_t *bsum, *bsum2;      // scan scratch (max(nf_in, nc)/256 + 1)
  int32_t *cp, *cu, *cn;              // compacted per-corner canonical value ids (old order)
  int32_t *opp, *vert;                // vert: vertex id per corner of the old-order table (position id, or n_pos + k for further fans of a non-manifold position)
  uint32_t *nmbits;                   // per position: several fans meet there (non-manifold): its corners' vertex ids are in vert[], all others' are cp[] (geo_vt)
  uint32_t extra_v, nseg[2]; uint32_t *vseam[2];   // ids handed out beyond n_pos; attribute segments; per-vertex 'an interior seam of attribute i touches it'
  uint8_t *vvis; int32_t *vval, *c2vm, *proc, *initc, *stack;      // vvis: vertex-visited bitmap of the edgebreaker walk when it is not in LDS
  uint8_t *evcnt;                      // topology-split events per symbol (auxiliary stream)
  int32_t *ev_src, *ev_spl; uint8_t *ev_edge;
  int32_t *rec[4]; uint8_t *symb, *ctx_of; int32_t *face_time;
  uint8_t *vopen_d[4]; int32_t *ring_d; uint32_t nverts_t[4];   // per vertex id: on a boundary, ring size; size of the id space per table (encoder: only vopen_d[0])
  uint32_t *ctx_all; uint32_t *ctx_sym[6]; uint32_t ctx_n[6];      // the six context streams back to back in ONE nf-entry array (k_eb_ctx counts, then scatters)
  uint32_t evcap, stcap;              // capacities of the split-event arrays / of the walkers' pending stacks (GEO_E_WS_OVERFLOW past them: retried with worst-case sizes)
  uint8_t *start_bits;
  // Decoder order stays VIRTUAL on the encode side (round 5): every table keeps the stored face order and the decoder's face
  // numbering appears only as tstart[] (decoder-order face -> code of its first corner in the stored tables), which the traversals'
  // component starts and the seam-bit order follow.  nopp / bvert / seam[] are the DECODE path's view (geom_decode.hip wires them
  // to its decoded tables, which are in decoder order by construction; tstart == nullptr there).
  int32_t *nopp, *bvert; uint8_t *seam[2];
  int32_t *tstart;
  uint8_t *fseam;                      // per stored face: bit k = the edge opposite corner k is a seam (or a boundary) of attribute slot 0, bit 3 + k: of slot 1
  uint8_t *sbpack;                     // per decoder-order face: seam bits it contributes: count (bits 0-1), slot-0 bits (2-4), slot-1 bits (5-7)
  uint8_t *seam_bits[2];
  int32_t *avert[2];                   // attribute vertex per corner, WRITTEN ONLY for corners of seam-touched vertices (vseam bit): all others keep vert[]
  int32_t base_hi;                     // 1: table 0 of the traversals is the walk's own record table (rec[1] == rec[0]); its visited flag is bit 127
  int32_t *order[3], *v2d[3]; uint8_t *t_vvis[3]; int32_t *t_stack[3];
  uint8_t *fvis, *t_fvis[3];          // face-visited bits (one per face) of the lane-per-walker kernels on per-face records (walk, three traversals)
  int32_t *P, *U, *O;                  // sequential connectivity only: quantised values per point
  uint16_t *qpos, *quv, *qnrm;         // quantised values BY VALUE ID (k_quant_ids): 4 x u16 per position (one 8-byte gather), 2 x u16 per uv / octahedral normal
  long long *fnorm;                    // per stored face: the un-normalised face normal (p1 - p0) x (p2 - p0) of its quantised positions (k_face_normals)
  uint32_t *sym_pos, *sym_uv, *sym_nrm;
  uint8_t *has_ori, *ori_val, *ori_c, *ori_bits, *flips;
  RansStream rs[GEO_NSTREAM];
  RabsStream rb[GEO_NRABS];
  uint8_t *arena; uint32_t arena_cap;
  uint32_t ecap;                                // capacity (entries) of the per-vertex / per-entry arrays (GEO_E_WS_OVERFLOW past it)
  uint64_t slab_cap;                            // bytes of the batch's packed output area (GEO_E_SLAB_FULL past it)
  uint8_t *ws_base; uint64_t ws_zero;          // zero-initialised head of this job's workspace (k_job_clear)
  const uint8_t *piece_ptr[GEO_MAXPIECES]; uint32_t piece_len[GEO_MAXPIECES], piece_off[GEO_MAXPIECES]; uint32_t n_pieces;
  uint32_t out_cap;
  uint8_t *out_pack; uint64_t out_pack_off;     // packed output area of the batch + this frame's offset in it (k_out_offsets)
};

#define GEO_INV (-1)
// device status codes the host reacts to: the compact workspace / the packed output area was too small for this frame
// (geo_encode_batch re-encodes such a frame alone with worst-case sizes)
#define GEO_E_WS_OVERFLOW (-50)
#define GEO_E_SLAB_FULL (-51)
#define GEO_E_DD_OVERFLOW (-52)        // a hash bin of the partitioned dedup holds more distinct values than its LDS table: retried with the hash-table dedup
// corner codes for the serial walkers: 4 * face + k, so that face = code >> 2 and records are indexed without a division
__device__ __forceinline__ int code_of_corner(int c) { return c < 0 ? -1 : (((c / 3) << 2) | (c % 3)); }
__device__ __host__ __forceinline__ uint32_t uvol_blocks_dev(uint32_t n) { return (n + 255u) / 256u; }
__device__ __host__ __forceinline__ int g_nxt(int c) { return (c % 3 == 2) ? c - 2 : c + 1; }
__device__ __host__ __forceinline__ int g_prv(int c) { return (c % 3 == 0) ? c + 2 : c - 1; }

__device__ __forceinline__ uint64_t g_mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
__device__ __forceinline__ uint32_t g_float_order(float f) {
  uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __host__ __forceinline__ float g_float_unorder(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; memcpy(&f, &u, 4); return f;
}

// seam-masked corner-table view
struct GTab { const int32_t *opp; const uint8_t *seam; };
__device__ __forceinline__ int gt_opp(const GTab &t, int c) { if (c < 0) return GEO_INV; if (t.seam && t.seam[c]) return GEO_INV; return t.opp[c]; }
__device__ __forceinline__ int gt_swl(const GTab &t, int c) { int o = gt_opp(t, g_nxt(c)); return o < 0 ? GEO_INV : g_nxt(o); }
__device__ __forceinline__ int gt_swr(const GTab &t, int c) { int o = gt_opp(t, g_prv(c)); return o < 0 ? GEO_INV : g_prv(o); }

// exact floor(sqrt(n)) for n < 2^63 (draco IntSqrt contract)
__device__ __forceinline__ uint64_t g_isqrt(uint64_t n) {
  if (n == 0) return 0;
  uint64_t r = (uint64_t)sqrt((double)n);
  while (r * r > n) r--;
  while ((r + 1) * (r + 1) <= n) r++;
  return r;
}

// ---- octahedral toolbox ----
struct GOct { int MAXQ, MAXV, CEN; };
__device__ __forceinline__ GOct g_oct(int q) { GOct t; t.MAXQ = (1 << q) - 1; t.MAXV = t.MAXQ - 1; t.CEN = t.MAXV / 2; return t; }
__device__ __forceinline__ int g_iabs(int x) { return x < 0 ? -x : x; }
__device__ __forceinline__ long long g_labs(long long x) { return x < 0 ? -x : x; }
__device__ inline void g_oct_canon(const GOct &t, int &s, int &tt) {
  if ((s == 0 && tt == 0) || (s == 0 && tt == t.MAXV) || (s == t.MAXV && tt == 0)) { s = t.MAXV; tt = t.MAXV; return; }
  if (s == 0 && tt > t.CEN) tt = t.CEN - (tt - t.CEN);
  else if (s == t.MAXV && tt < t.CEN) tt = t.CEN + (t.CEN - tt);
  else if (tt == t.MAXV && s < t.CEN) s = t.CEN + (t.CEN - s);
  else if (tt == 0 && s > t.CEN) s = t.CEN - (s - t.CEN);
}
__device__ inline void g_vec_to_oct(const GOct &t, const int v[3], int &s, int &tt) {
  if (v[0] >= 0) { s = v[1] + t.CEN; tt = v[2] + t.CEN; }
  else { s = v[1] < 0 ? g_iabs(v[2]) : t.MAXV - g_iabs(v[2]); tt = v[2] < 0 ? g_iabs(v[1]) : t.MAXV - g_iabs(v[1]); }
  g_oct_canon(t, s, tt);
}
__device__ inline void g_invert_diamond(const GOct &t, int &s, int &tt) {
  int ss, st;
  if (s >= 0 && tt >= 0) { ss = 1; st = 1; } else if (s <= 0 && tt <= 0) { ss = -1; st = -1; } else { ss = s > 0 ? 1 : -1; st = tt > 0 ? 1 : -1; }
  int cs = ss * t.CEN, ct = st * t.CEN, us = 2 * s - cs, ut = 2 * tt - ct;
  if (ss * st >= 0) { int tmp = us; us = -ut; ut = -tmp; } else { int tmp = us; us = ut; ut = tmp; }
  us += cs; ut += ct; s = us / 2; tt = ut / 2;
}
__device__ __forceinline__ int g_rot_count(int x, int y) {
  if (x == 0) return y == 0 ? 0 : (y > 0 ? 3 : 1);
  if (x > 0) return y >= 0 ? 2 : 1;
  return y <= 0 ? 0 : 3;
}
__device__ __forceinline__ void g_rot(int &x, int &y, int c) {
  int X = x, Y = y;
  if (c == 1) { x = Y; y = -X; } else if (c == 2) { x = -X; y = -Y; } else if (c == 3) { x = -Y; y = X; }
}
__device__ __forceinline__ int g_modmax(const GOct &t, int x) { if (x > t.CEN) return x - t.MAXQ; if (x < -t.CEN) return x + t.MAXQ; return x; }
__device__ inline void g_oct_corr(const GOct &t, const int orig[2], const int pred[2], int corr[2]) {
  int os = orig[0] - t.CEN, ot = orig[1] - t.CEN, ps = pred[0] - t.CEN, pt = pred[1] - t.CEN;
  if (g_iabs(ps) + g_iabs(pt) > t.CEN) { g_invert_diamond(t, os, ot); g_invert_diamond(t, ps, pt); }
  bool bl = (ps == 0 && pt == 0) || (ps < 0 && pt <= 0);
  if (!bl) { int rc = g_rot_count(ps, pt); g_rot(os, ot, rc); g_rot(ps, pt, rc); }
  corr[0] = os - ps; corr[1] = ot - pt;
  if (corr[0] < 0) corr[0] += t.MAXQ;
  if (corr[1] < 0) corr[1] += t.MAXQ;
}

__device__ __forceinline__ uint32_t g_sym_of(int v) { return v >= 0 ? ((uint32_t)v << 1) : ((((uint32_t)(-(v + 1))) << 1) | 1u); }
__device__ __forceinline__ int g_wrap_corr(int lo, int hi, int orig, long long pred) {
  int max_dif = 1 + hi - lo, max_corr = max_dif / 2, min_corr = -max_corr;
  if ((max_dif & 1) == 0) max_corr -= 1;
  int p = pred < lo ? lo : (pred > hi ? hi : (int)pred);
  int c = orig - p;
  if (c < min_corr) c += max_dif; else if (c > max_corr) c -= max_dif;
  return c;
}

__device__ __forceinline__ uint32_t g_put_varint(uint8_t *p, uint64_t v) {
  uint32_t n = 0; while (v >= 0x80) { p[n++] = (uint8_t)(v | 0x80); v >>= 7; } p[n++] = (uint8_t)v; return n;
}
__device__ __forceinline__ uint32_t g_varint_len(uint64_t v) { uint32_t n = 1; while (v >= 0x80) { n++; v >>= 7; } return n; }
