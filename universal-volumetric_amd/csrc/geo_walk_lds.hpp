// geo_walk_lds.hpp - K4: walker records and the LDS (wave-per-walker) forms of the edgebreaker walk.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
// ------------------------------------------------------------------------------------------------
// K4: valence edgebreaker — split into
//   k_pack0        (parallel)  per-corner records {vertex<<1|open, right, left[, opposite]} (8 or 16 bytes, RecOps) indexed by corner code
//                              4*face+k, so a walker step is ONE load and no division / select
//   k_eb_walk      (serial)    MeshEdgebreakerEncoderImpl::EncodeConnectivity traversal only: symbols + processed corners;
//                              visited faces / vertices are bitmaps in LDS (k_face_time inverts proc[] afterwards)
//   k_eb_events    (parallel)  topology-split events from (symbol, neighbour symbol) pairs, order-preserving compaction
//   k_eb_valence   (1 wave)    MeshEdgebreakerTraversalValenceEncoder bookkeeping replayed over the known symbol
//                              sequence: runs between split symbols are resolved by all 64 lanes at once
//   k_eb_ctx       (1 wave)    ballot-ordered scatter of the symbols into the 6 valence-context streams
// (SURVEY A.3 / A.10).  The serial kernels run one frame per workgroup; a batch keeps that many CUs busy.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dense_table_live(const GeoJob &J, int which) { const int ai = which >= 2 ? which - 2 : 0; return !(which >= 2 && (ai >= J.nad || !J.interior_seams[ai])); }
// the three records of face f (r[k] = opposite corner of corner k, vc[k] = vertex << 1 | open) in either format
// r8: 0 = 16 bytes per corner, 1 = 8 bytes per corner, 2 = ONE 16-byte record per FACE (f16_*, the lane-per-walker kernels)
__device__ __forceinline__ void pack_face_records(int32_t *rec, uint32_t f, const int vc[3], const int r[3], int r8) {
  if (r8 == 2) {
    // {vertex << 1 | open} x 3 in the low 64 bits (bit 63: the walker's face-visited flag), the opposite corner codes x 3 in the high
    // 64 bits, 21-bit fields: what the three 8-byte corner records hold (each opposite twice) in half the bytes, and the flag in it
    uint64_t lo = 0, hi = 0;
    for (int k = 0; k < 3; k++) { lo |= (uint64_t)((uint32_t)vc[k] & 0x1fffffu) << (21 * k); hi |= (uint64_t)((uint32_t)code_of_corner(r[k]) & 0x1fffffu) << (21 * k); }
    reinterpret_cast<uint4 *>(rec)[f] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
  } else if (r8) {
    uint4 *dst = reinterpret_cast<uint4 *>(rec) + 2 * (size_t)f;      // the face's 32-byte block as two 16-byte stores
    uint2 q[3];
    for (int k = 0; k < 3; k++) {
      const uint32_t R = (uint32_t)code_of_corner(r[(k + 1) % 3]) & 0x1fffffu, L = (uint32_t)code_of_corner(r[(k + 2) % 3]) & 0x1fffffu;
      q[k] = make_uint2(((uint32_t)vc[k] & 0x1fffffu) | (R << 21), (R >> 11) | (L << 10));
    }
    dst[0] = make_uint4(q[0].x, q[0].y, q[1].x, q[1].y);
    dst[1] = make_uint4(q[2].x, q[2].y, 0u, 0u);    // 4th slot of the face's block: "face visited" flag of the lane-per-walker kernels
  } else {
    int4 *dst = reinterpret_cast<int4 *>(rec) + 4 * (size_t)f;
    for (int k = 0; k < 3; k++) dst[k] = make_int4(vc[k], code_of_corner(r[(k + 1) % 3]), code_of_corner(r[(k + 2) % 3]), code_of_corner(r[k]));
    dst[3] = make_int4(0, 0, 0, 0);
  }
}
// per-face records carry one more record at index nf: the DUMMY face (both visited flags set, no neighbours) that k_traverse_wave_f16
// stands on while it pops, and that "no neighbour" clamps to
__device__ __forceinline__ void pack_dummy_record(int32_t *rec, uint32_t nf, int r8) { if (r8 == 2) reinterpret_cast<uint4 *>(rec)[nf] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu); }
// decode path (geom_decode.hip prepares vert / vopen_d itself): which: 1 new base, 2/3 attribute tables (DFS)
__global__ void __launch_bounds__(UVOL_BLOCK) k_pack_faces(GeoJob *jobs, int which, int r8) {
  JOB_OR_RETURN;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f >= J.nf) return;
  const int ai = which >= 2 ? which - 2 : 0;
  if (which >= 2 && (ai >= J.nad || !J.interior_seams[ai])) return;
  if (f == 0) pack_dummy_record(J.rec[which], J.nf, r8);
  const int32_t *opp = which == 0 ? J.opp : J.nopp;
  const uint8_t *seam = which >= 2 ? J.seam[ai] : nullptr;
  const int32_t *vert = which == 0 ? J.vert : (which == 1 ? J.bvert : J.avert[ai]);
  const uint8_t *vopen = J.vopen_d[which];
  int r[3], vc[3];
  for (int k = 0; k < 3; k++) {
    const int c = 3 * (int)f + k;
    r[k] = (seam && seam[c]) ? GEO_INV : opp[c];
    const int v = vert[c];
    vc[k] = (v << 1) | (vopen[v] ? 1 : 0);
  }
  pack_face_records(J.rec[which], f, vc, r, r8);
  if (which == 0) J.face_time[f] = -1;            // faces that start a component without a symbol keep -1 (see k_face_time)
}
// encoder, table 0 (old order): records for the edgebreaker walk; also publishes the size of the vertex id space
__global__ void __launch_bounds__(UVOL_BLOCK) k_pack0(GeoJob *jobs, int r8) {
  JOB_OR_RETURN;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f == 0) { const uint32_t tot = J.n_pos + J.extra_v; J.nverts_t[0] = tot; J.nverts_t[1] = tot; if (tot > J.ecap) J.status = GEO_E_WS_OVERFLOW; }
  if (f >= J.nf) return;
  int r[3], vc[3];
  for (int k = 0; k < 3; k++) {
    const int c = 3 * (int)f + k;
    r[k] = J.opp[c];
    const uint32_t v = (uint32_t)geo_vt(J)[c];
    vc[k] = (int)((v << 1) | ((v < J.ecap && J.vopen_d[0][v]) ? 1u : 0u));
  }
  pack_face_records(J.rec[0], f, vc, r, r8);
  if (f == 0) pack_dummy_record(J.rec[0], J.nf, r8);
  J.face_time[f] = -1;                            // faces that start a component without a symbol keep -1 (see k_face_time)
}
// encoder, tables first .. 3 (table = first + blockIdx.z), in the STORED face order like table 0: the attribute tables that have interior
// seams (2, 3) and - only when the walkers do not read one 16-byte record per face - a copy of the base table (1).  With per-face
// records table 1 IS table 0 (GeoJob::base_hi: the traversal marks its faces in bit 127, the walk used bit 63).
// An attribute vertex an interior seam does not touch keeps its base id and open flag; the segments of the others are open.
__global__ void __launch_bounds__(UVOL_BLOCK) k_pack_tabs(GeoJob *jobs, int r8, int first) {
  JOB_OR_RETURN;
  const int which = first + (int)blockIdx.z;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f >= J.nf || !dense_table_live(J, which)) return;
  if (f == 0) pack_dummy_record(J.rec[which], J.nf, r8);
  const int ai = which >= 2 ? which - 2 : -1;
  const uint32_t nbase = J.nverts_t[0], fs = ai >= 0 ? (uint32_t)J.fseam[f] >> (3 * ai) : 0u;
  const uvol_s3 o3 = *reinterpret_cast<const uvol_s3 *>(J.opp + 3 * (size_t)f), v3 = *reinterpret_cast<const uvol_s3 *>(geo_vt(J) + 3 * (size_t)f);
  const int oo[3] = { o3.x, o3.y, o3.z }, vv[3] = { v3.x, v3.y, v3.z };
  int r[3], vc[3];
  for (int k = 0; k < 3; k++) {
    r[k] = ((fs >> k) & 1u) ? GEO_INV : oo[k];
    uint32_t v = (uint32_t)vv[k];
    if (ai >= 0 && ((J.vseam[ai][v >> 5] >> (v & 31)) & 1u)) v = (uint32_t)J.avert[ai][3 * f + k];
    vc[k] = (int)((v << 1) | ((v >= nbase || J.vopen_d[0][v]) ? 1u : 0u));
  }
  pack_face_records(J.rec[which], f, vc, r, r8);
}

// typed-pointer helpers for the one-lane walkers (P = UVOL_G / UVOL_L pointer)
#ifdef HIPEMU
typedef int4 uvol_i4;
#else
typedef int uvol_i4 __attribute__((ext_vector_type(4)));      // loadable through an address-space-qualified pointer
#endif
// the three live words {vertex, right, left} of a corner record as ONE 12-byte load: a prefetched 16-byte load would leave
// its dead 4th register free for the allocator to reuse at once, which forces a wait right behind the load
#ifdef HIPEMU
struct uvol_i3 { int x, y, z; };
template <typename P> __device__ __forceinline__ uvol_i3 rec3(P rec, int code) { const uvol_i4 q = rec[code]; uvol_i3 r; r.x = q.x; r.y = q.y; r.z = q.z; return r; }
#else
typedef int uvol_i3 __attribute__((ext_vector_type(3)));
template <typename P> __device__ __forceinline__ uvol_i3 rec3(P rec, int code) { return *(UVOL_G(const uvol_i3))(rec + code); }
#endif
// bitmap words: LDS pointers read with ds_read; global pointers with a workgroup-scope atomic load (sc0: not served from a
// possibly stale per-CU L1 line - the bits are set with atomic ORs performed in L2 - but, unlike the device-scope load used
// before, served by this XCD's L2 instead of the memory-side cache: one walker wave is the only reader and writer of its bitmap)
__device__ __forceinline__ uint32_t pword(UVOL_L(uint32_t) w, int k) { return w[k]; }
#ifndef HIPEMU
__device__ __forceinline__ uint32_t pword(UVOL_G(uint32_t) w, int k) { return __hip_atomic_load(&w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
template <typename P> __device__ __forceinline__ bool pbit_get(P w, int i) { return (pword(w, i >> 5) >> (i & 31)) & 1u; }
template <typename P> __device__ __forceinline__ void pbit_set(P w, int i) { UVOL_OR_NORET(&w[i >> 5], 1u << (i & 31)); }   // fire and forget
__device__ __forceinline__ int code_nxt(int x) { return (x & 3) == 2 ? x - 2 : x + 1; }
__device__ __forceinline__ int code_prv(int x) { return (x & 3) == 0 ? x + 2 : x - 1; }
__device__ __forceinline__ int corner_of_code(int x) { return 3 * (x >> 2) + (x & 3); }

// Corner records in two formats.  R8 = false: 16 bytes {vertex<<1|open, right, left, opposite} (any mesh size).  R8 = true:
// 8 bytes, three 21-bit fields {vertex<<1|open : 0..20, right : 21..41, left : 42..62} (codes and ids < 2^20, -1 = all
// ones), used whenever the batch allows it: a 128-byte line then holds the records of four faces instead of two, so more of
// a walker's dependent loads hit a line a neighbouring face already brought in, and the walkers and k_pack move half the
// bytes.  The opposite corner is not stored: opposite(k) = right field of the record of corner (k + 2) % 3.
#ifdef HIPEMU
struct uvol_u2 { uint32_t x, y; };
#else
typedef uint32_t uvol_u2 __attribute__((ext_vector_type(2)));
#endif
__device__ __forceinline__ void rec8_dec(uint32_t lo, uint32_t hi, int &vi, int &rc, int &lc) {
  vi = (int)(lo & 0x1fffffu);
  rc = (int)(((lo >> 21) | (hi << 11)) << 11) >> 11;
  lc = (int)(hi << 1) >> 11;
}
template <bool R8> struct RecOps;
template <> struct RecOps<false> {
  typedef UVOL_G(const uvol_i4) Ptr; typedef uvol_i3 Pre;
  static __device__ __forceinline__ Ptr ptr(const int32_t *p) { return UVOL_TO_G(const uvol_i4, reinterpret_cast<const uvol_i4 *>(p)); }
  static __device__ __forceinline__ void get(Ptr rec, int code, int &vi, int &rc, int &lc) { const uvol_i4 q = rec[code]; vi = q.x; rc = q.y; lc = q.z; }
  static __device__ __forceinline__ Pre pre(Ptr rec, int code) { return rec3(rec, code); }
  static __device__ __forceinline__ void take(const Pre &p, int &vi, int &rc, int &lc) { vi = UVOL_READFIRST(p.x); const int l_ = UVOL_READFIRST(p.z); rc = UVOL_READFIRST(p.y); lc = l_; }
};
template <> struct RecOps<true> {
  typedef UVOL_G(const uvol_u2) Ptr; typedef uvol_u2 Pre;
  static __device__ __forceinline__ Ptr ptr(const int32_t *p) { return UVOL_TO_G(const uvol_u2, reinterpret_cast<const uvol_u2 *>(p)); }
  static __device__ __forceinline__ void get(Ptr rec, int code, int &vi, int &rc, int &lc) { const uvol_u2 q = rec[code]; rec8_dec(q.x, q.y, vi, rc, lc); }
  static __device__ __forceinline__ Pre pre(Ptr rec, int code) { return rec[code]; }
  static __device__ __forceinline__ void take(const Pre &p, int &vi, int &rc, int &lc) { const uint32_t lo = (uint32_t)UVOL_READFIRST(p.x), hi = (uint32_t)UVOL_READFIRST(p.y); rec8_dec(lo, hi, vi, rc, lc); }
};

// Output staging of the one-lane LDS walkers.  gfx950 has ONE counter (vmcnt) for loads and stores and the compiler treats a queue
// that holds both as unordered: with a store outstanding, the wait for the record the next step needs becomes vmcnt(0) and also
// waits for the acknowledgement of the proc[] / symb[] (order[]) stores issued a moment ago - a second memory round trip per face
// on top of the record load.  The walkers therefore write their output streams to LDS (lgkmcnt) and flush WALK_STG entries at a
// time with 16-byte stores: one store acknowledgement per WALK_STG faces instead of one per face.
#define WALK_STG 256                                    // staged entries (multiple of 16)
#define WALK_STG_DWORDS (WALK_STG + WALK_STG / 4)       // int32 entries + one byte per entry
struct WalkStage {
  UVOL_L(int32_t) w; UVOL_L(uint8_t) b;
  __device__ __forceinline__ void init(UVOL_L(uint32_t) lds) { w = (UVOL_L(int32_t))lds; b = (UVOL_L(uint8_t))(lds + WALK_STG); }
  // entries [n - WALK_STG, n) of the streams leave when n reaches a multiple of WALK_STG (16-byte aligned: arrays are 256-byte aligned)
  __device__ __forceinline__ void flush_words(UVOL_G(int32_t) dst, int n) {
    UVOL_G(uvol_i4) d = (UVOL_G(uvol_i4))(dst + (n - WALK_STG)); UVOL_L(const uvol_i4) s = (UVOL_L(const uvol_i4))w;
#pragma unroll 8
    for (int i = 0; i < WALK_STG / 4; i++) d[i] = s[i];
  }
  __device__ __forceinline__ void flush_bytes(UVOL_G(uint8_t) dst, int n) {
    UVOL_G(uvol_i4) d = (UVOL_G(uvol_i4))(dst + (n - WALK_STG)); UVOL_L(const uvol_i4) s = (UVOL_L(const uvol_i4))b;
#pragma unroll 8
    for (int i = 0; i < WALK_STG / 16; i++) d[i] = s[i];
  }
  __device__ __forceinline__ void tail_words(UVOL_G(int32_t) dst, int n) { for (int i = n & ~(WALK_STG - 1); i < n; i++) dst[i] = w[i & (WALK_STG - 1)]; }
  __device__ __forceinline__ void tail_bytes(UVOL_G(uint8_t) dst, int n) { for (int i = n & ~(WALK_STG - 1); i < n; i++) dst[i] = b[i & (WALK_STG - 1)]; }
};
// Edgebreaker walk, one lane per frame: typed pointers (global_* / ds_* instructions, exactly counted waits), no scatter
// stores (face_time is rebuilt from proc[] by k_face_time).  Per face: ONE 8- or 16-byte record read from HBM — the dependent
// access that bounds the walk —, one sequential proc/symb store pair, a fire-and-forget ds_or for the face bit and one
// LDS round trip for the vertex / neighbour bits.  Corners are carried as codes (4 * face + k).
template <bool R8, typename FB, typename VB>
__device__ __forceinline__ void eb_walk_lane0(GeoJob &J, FB fbits, VB vbits, UVOL_L(uint32_t) stg_lds) {
  typedef RecOps<R8> RO;
  const int nf = (int)J.nf;
  const typename RO::Ptr rec = RO::ptr(J.rec[0]);
  UVOL_G(int32_t) proc = UVOL_TO_G(int32_t, J.proc); UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.stack);
  UVOL_G(int32_t) initc = UVOL_TO_G(int32_t, J.initc);
  UVOL_G(uint8_t) symb = UVOL_TO_G(uint8_t, J.symb); UVOL_G(uint8_t) start_bits = UVOL_TO_G(uint8_t, J.start_bits);
  const int dz = UVOL_LANE_ZERO();
  WalkStage stg; stg.init(stg_lds);
  int nproc = 0, ninit = 0, nstart = 0, nsplit = 0;
#define W_EMIT(SYM) do { stg.b[nproc & (WALK_STG - 1)] = (uint8_t)(SYM); nproc++; if ((nproc & (WALK_STG - 1)) == 0) { stg.flush_words(proc, nproc); stg.flush_bytes(symb, nproc); } } while (0)
  enum { T_C = 0, T_S = 1, T_L = 3, T_R = 5, T_E = 7 };
  const bool rl = J.relabel != 0; UVOL_G(const int32_t) s_of_o = UVOL_TO_G(const int32_t, J.s_of_o);
  for (int fo = 0; fo < nf; fo++) {
    // component starts, in the ORIGINAL face order (a relabelled frame maps it to the stored face); fully visited words of the
    // face bitmap are skipped 32 faces at a time where stored order = original order
    int f0 = fo;
    if (rl) { if (nproc + ninit >= nf) break; f0 = s_of_o[fo]; }
    else if ((fo & 31) == 0) { while (fo + 32 <= nf && pword(fbits, fo >> 5) == 0xffffffffu) fo += 32; if (fo >= nf) break; f0 = fo; }
    if (pbit_get(fbits, f0)) continue;
    int v0[3], r0_[3], l0_[3];
    for (int k = 0; k < 3; k++) RO::get(rec, 4 * f0 + k, v0[k], r0_[k], l0_[k]);
    const int o0[3] = { r0_[2], r0_[0], r0_[1] };                       // opposite(k) = right field of corner (k + 2) % 3
    int interior = 1, start = 4 * f0;
    for (int k = 0; k < 3; k++) {
      if (o0[k] < 0) { interior = 0; start = 4 * f0 + k; break; }
      if (v0[k] & 1) {                // boundary vertex: swing right to the boundary edge
        int ci = 4 * f0 + k, rc = ci;
        while (rc >= 0) { ci = rc; int v_, r_, o; RO::get(rec, rc, v_, r_, o); rc = o < 0 ? -1 : code_prv(o); }
        interior = 0; start = code_prv(ci); break;
      }
    }
    start_bits[nstart] = (uint8_t)interior;
    nstart++;
    int from;
    if (interior) {
      pbit_set(vbits, v0[0] >> 1); pbit_set(vbits, v0[1] >> 1); pbit_set(vbits, v0[2] >> 1);
      pbit_set(fbits, f0);
      initc[ninit] = 3 * f0 + 1;
      ninit++;
      from = o0[1];
      if (from < 0 || pbit_get(fbits, from >> 2)) continue;
    } else from = start;
    int sp = 0;
    stack[sp] = from;
    sp++;
    int top = from;                                   // value at stack[sp-1] when known without a load
    bool top_known = true;
    while (sp > 0) {
      int x = top_known ? top : stack[sp - 1];
      top_known = false;
      if (x < 0 || pbit_get(fbits, x >> 2)) { sp--; continue; }
      int vi, rcn, lcn;
      RO::get(rec, x, vi, rcn, lcn);
      for (;;) {
        const int face = x >> 2;
        // both records this step can move to are requested now and taken (readfirstlane) only by the branch that goes there
        const typename RO::Pre pR = RO::pre(rec, (rcn < 0 ? x : rcn) + dz), pL = RO::pre(rec, (lcn < 0 ? x : lcn) + dz);
        stg.w[nproc & (WALK_STG - 1)] = 3 * face + (x & 3);
        pbit_set(fbits, face);
        const int v = vi >> 1;
        // the three bitmap words this step can need, read together (one LDS round trip)
        const uint32_t vw_ = pword(vbits, v >> 5);
        const uint32_t rw_ = rcn < 0 ? 0xffffffffu : pword(fbits, rcn >> 7), lw_ = lcn < 0 ? 0xffffffffu : pword(fbits, lcn >> 7);
#define W_GO_R() do { x = rcn; RO::take(pR, vi, rcn, lcn); } while (0)
#define W_GO_L() do { x = lcn; RO::take(pL, vi, rcn, lcn); } while (0)
        if (!((vw_ >> (v & 31)) & 1u)) {
          pbit_set(vbits, v);
          if (!(vi & 1)) { W_EMIT(T_C); W_GO_R(); continue; }
        }
        const bool rvis = ((rw_ >> ((rcn >> 2) & 31)) & 1u) != 0, lvis = ((lw_ >> ((lcn >> 2) & 31)) & 1u) != 0;
        const int sym = rvis ? (lvis ? T_E : T_R) : (lvis ? T_L : T_S);
        W_EMIT(sym);
        if (sym == T_E) { sp--; break; }
        if (sym == T_R) { W_GO_L(); continue; }
        if (sym == T_L) { W_GO_R(); continue; }
        nsplit++;
        stack[sp - 1] = lcn; stack[sp] = rcn;
        sp++; top = rcn; top_known = true;
        break;
#undef W_GO_R
#undef W_GO_L
      }
    }
  }
#undef W_EMIT
  stg.tail_words(proc, nproc); stg.tail_bytes(symb, nproc);
  J.nsym = nproc; J.nsplit = nsplit; J.nstart = nstart; J.ninit = ninit;
  if (nproc + ninit != nf) J.status = -10;
  J.rb[0].n = (uint32_t)nstart;
  uint32_t z = 0; for (int i = 0; i < nstart; i++) z += J.start_bits[i] == 0;
  J.rb[0].zeros = z;
}


// ------------------------------------------------------------------------------------------------
// Cooperative-lane forms of the LDS walkers (the default whenever the bitmaps are in LDS).  A one-lane walk is bound by
// instruction issue, not by memory: a single wave issues about one instruction per 5 cycles, and a step of eb_walk_lane0 is
// ~100 instructions (two address computations and loads, three LDS reads behind branches, lane elections around the LDS
// atomics, scalar bookkeeping) = ~210 ns on top of the ~160 ns its dependent load costs (tools/latbench/seqbench).  Here the
// per-candidate work of a step is ONE vector instruction each: lane 0 handles the right neighbour, lane 1 the left one, every
// other lane the tip vertex - one load fetches both neighbours' records, one ds_read their two face-visited words and the
// vertex-visited word, one ballot turns the three tests into a scalar mask; the record the walk moves to is picked with
// v_readlane (lane select in an SGPR), so the step has no divergent branch and ~45 instructions.  The current face's bit is set with
// a plain LDS write (its word is known: a candidate lane read it one step earlier, or the pop test just did), outputs are staged in
// LDS (WalkStage) and flushed by all 64 lanes.  A second wave of the workgroup reads the walker's position from LDS and touches
// the 128-byte lines of the record table around it, so that the walker's loads hit in this CU's L1 / this XCD's L2 instead of
// paying an HBM miss per new line (seqbench: 157 -> 96 ns per dependent load on a strip-ordered table).
// Results are identical to eb_walk_lane0 / traverse_lane0 (same traversal, same output arrays).
// ------------------------------------------------------------------------------------------------
#define WALK_PUB_DWORDS 8                                // [0] walker position (corner code), [1] done flag
#define WALK_PF_LINES 64                                 // 128-byte lines the helper wave keeps touched around the walker
template <bool R8> struct CoopRec;
template <> struct CoopRec<true> {
  // two v_readlane, then scalar 64-bit shifts (written with 32-bit pieces the compiler moved the funnel shift back to the VALU)
  static __device__ __forceinline__ void take(const uvol_u2 &p, int sel, int &vi, int &rc, int &lc) {
    const unsigned long long q = ((unsigned long long)UVOL_READLANE(p.y, sel) << 32) | (unsigned long long)UVOL_READLANE(p.x, sel);
    vi = (int)((uint32_t)q & 0x1fffffu); rc = (int)((long long)(q << 22) >> 43); lc = (int)((long long)(q << 1) >> 43);
  }
};
template <> struct CoopRec<false> {
  static __device__ __forceinline__ void take(const uvol_i3 &p, int sel, int &vi, int &rc, int &lc) { vi = (int)UVOL_READLANE(p.x, sel); rc = (int)UVOL_READLANE(p.y, sel); lc = (int)UVOL_READLANE(p.z, sel); }
};
// record of `code` as wave-uniform scalars (every lane loads the same address: one request)
template <bool R8> __device__ __forceinline__ void coop_get(typename RecOps<R8>::Ptr rec, int code, int &vi, int &rc, int &lc) {
  int a, b, c; RecOps<R8>::get(rec, code, a, b, c); vi = UVOL_READFIRST(a); rc = UVOL_READFIRST(b); lc = UVOL_READFIRST(c);
}
// helper wave: keeps WALK_PF_LINES lines of the record table around the walker's published position touched
__device__ __forceinline__ void walk_prefetch_wave(const int32_t *rec_base, uint32_t rec_bytes, UVOL_L(uint32_t) pub, int shift /* corner code -> 128-byte line */) {
#ifndef HIPEMU
  const int lane = (int)(threadIdx.x & 63), nlines = (int)(rec_bytes >> 7);
  UVOL_G(const uint32_t) r = UVOL_TO_G(const uint32_t, reinterpret_cast<const uint32_t *>(rec_base));
  int base = -(1 << 30); uint32_t acc = 0;
  while (!__hip_atomic_load(&pub[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
    const int c = (int)__hip_atomic_load(&pub[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> shift;
    if (c < base + WALK_PF_LINES / 4 || c >= base + (3 * WALK_PF_LINES) / 4) {
      base = c - WALK_PF_LINES / 4;
      const int line = base + lane;
      if (line >= 0 && line < nlines) acc += r[32 * (size_t)line];
    }
    __builtin_amdgcn_s_sleep(2);
  }
  if (acc == 0x9e3779b9u) pub[2] = acc;               // keeps the loads alive
#endif
}

template <bool R8>
__device__ __forceinline__ void eb_walk_coop(GeoJob &J, UVOL_L(uint32_t) lds, uint32_t fw, UVOL_L(uint32_t) pub, int pf) {
  typedef RecOps<R8> RO;
  const int lane = (int)(threadIdx.x & 63);
  const bool cl = lane < 2;                              // candidate lanes: 0 = right neighbour, 1 = left neighbour; the others: tip vertex
  const int nf = (int)J.nf;
  const typename RO::Ptr rec = RO::ptr(J.rec[0]);
  UVOL_G(int32_t) proc = UVOL_TO_G(int32_t, J.proc); UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.stack);
  UVOL_G(int32_t) initc = UVOL_TO_G(int32_t, J.initc);
  UVOL_G(uint8_t) symb = UVOL_TO_G(uint8_t, J.symb); UVOL_G(uint8_t) start_bits = UVOL_TO_G(uint8_t, J.start_bits);
  UVOL_L(uint32_t) dummy = pub + 4 + (lane & 1);         // where the candidate lanes put the word the vertex lanes write back
  uint32_t pv = 0, sv = 0;                               // output staging: lane k = entry (nproc & ~63) + k of proc[] / symb[]
  int nproc = 0, ninit = 0, nstart = 0, nsplit = 0;
#define C_FWORD(k) ((uint32_t)UVOL_BCAST0(lds[k]))
  const bool rl = J.relabel != 0; UVOL_G(const int32_t) s_of_o = UVOL_TO_G(const int32_t, J.s_of_o);
  for (int fo = 0; fo < nf; fo++) {
    int f0 = fo;                                         // component starts follow the ORIGINAL face order (see eb_walk_lane0)
    if (rl) { if (nproc + ninit >= nf) break; f0 = UVOL_READFIRST(s_of_o[fo]); }
    else if ((fo & 31) == 0) { while (fo + 32 <= nf && C_FWORD(fo >> 5) == 0xffffffffu) fo += 32; if (fo >= nf) break; f0 = fo; }
    if ((C_FWORD(f0 >> 5) >> (f0 & 31)) & 1u) continue;
    int v0[3], r0_[3], l0_[3];
    for (int k = 0; k < 3; k++) coop_get<R8>(rec, 4 * f0 + k, v0[k], r0_[k], l0_[k]);
    const int o0[3] = { r0_[2], r0_[0], r0_[1] };
    int interior = 1, start = 4 * f0;
    for (int k = 0; k < 3; k++) {
      if (o0[k] < 0) { interior = 0; start = 4 * f0 + k; break; }
      if (v0[k] & 1) {
        int ci = 4 * f0 + k, rc = ci;
        while (rc >= 0) { ci = rc; int v_, r_, o; coop_get<R8>(rec, rc, v_, r_, o); rc = o < 0 ? -1 : code_prv(o); }
        interior = 0; start = code_prv(ci); break;
      }
    }
    if (lane == 0) start_bits[nstart] = (uint8_t)interior;
    nstart++;
    int from;
    if (interior) {
      for (int k = 0; k < 3; k++) { const int v = v0[k] >> 1; lds[fw + (v >> 5)] = (uint32_t)UVOL_BCAST0(lds[fw + (v >> 5)]) | (1u << (v & 31)); }
      lds[f0 >> 5] = C_FWORD(f0 >> 5) | (1u << (f0 & 31));
      if (lane == 0) initc[ninit] = 3 * f0 + 1;
      ninit++;
      from = o0[1];
      if (from < 0 || ((C_FWORD(from >> 7) >> ((from >> 2) & 31)) & 1u)) continue;
    } else from = start;
    int sp = 0;
    if (lane == 0) stack[sp] = from;
    sp++;
    int top = from; bool top_known = true;
    while (sp > 0) {
      int x;
      if (top_known) x = top; else { UVOL_WAVE_FENCE(); x = UVOL_BCAST0(stack[sp - 1]); }     // lane 0's own earlier store
      top_known = false;
      if (x < 0) { sp--; continue; }
      uint32_t xw = C_FWORD(x >> 7);                     // face-visited word of x's face
      if ((xw >> ((x >> 2) & 31)) & 1u) { sp--; continue; }
      int vi, rcn, lcn;
      coop_get<R8>(rec, x, vi, rcn, lcn);
      // One step = straight-line code with ONE taken branch (the back edge): a lone wave pays ~40 cycles of instruction fetch per
      // taken branch, so the common symbols (C, R, L) are resolved with scalar selects; S / E (a few % of the steps) and the
      // write-out of the staged outputs (every 64th step) leave the line.
      for (;;) {
        const int face = x >> 2;
        const int cand = lane == 0 ? rcn : lcn; const bool cvalid = cand >= 0;
        const int ccode = cvalid ? cand : x;
        const typename RO::Pre pre = RO::pre(rec, ccode);                        // lanes 0 / 1: the two records this step can move to
        if (pf) pub[0] = (uint32_t)x;                                            // for the prefetch wave
        lds[face >> 5] = xw | (1u << (face & 31));                               // face visited (plain write: xw is current)
        pv = UVOL_WRITELANE(3 * face + (x & 3), nproc & 63, pv);
        const int v = vi >> 1;
        const uint32_t widx = cl ? (uint32_t)ccode >> 7 : fw + (uint32_t)(v >> 5);
        const uint32_t sh = cl ? ((uint32_t)cand >> 2) & 31u : (uint32_t)v & 31u;
        const uint32_t word = lds[widx];
        const bool hit = ((word >> sh) & 1u) != 0 || (cl && !cvalid);
        const uint32_t m = (uint32_t)__ballot(hit) & 7u;                         // bit 0: right visited, 1: left visited, 2: tip vertex visited
        (cl ? dummy : lds + widx)[0] = word | (1u << sh);                        // the tip's bit (already set when it was visited)
        const bool ccase = (((m >> 2) | (uint32_t)vi) & 1u) == 0;                // tip unvisited and not on a boundary: C
        const uint32_t sym = ccase ? 0u : 1u + (m & 2u) + ((m & 1u) << 2);       // S = 1, L = 3 (left visited), R = 5 (right visited), E = 7
        sv = UVOL_WRITELANE(sym, nproc & 63, sv);
        nproc++;
        if (__builtin_expect((nproc & 63) == 0, 0)) { proc[nproc - 64 + lane] = (int32_t)pv; symb[nproc - 64 + lane] = (uint8_t)sv; }
        if (__builtin_expect((0x82u >> sym) & 1u, 0)) {                            // E (7) or S (1): the run of C / R / L steps ends
          if (sym == 7u) { sp--; break; }
          nsplit++;
          if (lane == 0) { stack[sp - 1] = lcn; stack[sp] = rcn; }
          sp++; top = rcn; top_known = true;
          break;
        }
        const int sel = (int)(sym >> 2);                                          // R (5): the walk goes left; C (0) and L (3): right
        x = sel ? lcn : rcn;
        xw = UVOL_READLANE(word, sel);
        CoopRec<R8>::take(pre, sel, vi, rcn, lcn);
      }
    }
  }
  if (lane < (nproc & 63)) { proc[(nproc & ~63) + lane] = (int32_t)pv; symb[(nproc & ~63) + lane] = (uint8_t)sv; }
  if (lane == 0) {
    J.nsym = nproc; J.nsplit = nsplit; J.nstart = nstart; J.ninit = ninit;
    if (nproc + ninit != nf) J.status = -10;
    J.rb[0].n = (uint32_t)nstart;
    UVOL_WAVE_FENCE();
    uint32_t z = 0; for (int i = 0; i < nstart; i++) z += J.start_bits[i] == 0;
    J.rb[0].zeros = z;
  }
}
#undef C_FWORD

// LDS: [face bits, fw words][vertex bits, vcap_words]; vcap_words is sized by the host from the input attribute counts and
// the LDS slot (a table with more vertices keeps its vertex bitmap in global memory).  A mesh whose face bitmap does not fit
// LDS is walked by the lane-per-walker kernels below (nothing in LDS).
template <bool R8>
__global__ void __launch_bounds__(128) k_eb_walk(GeoJob *jobs, int vcap_words, int pf) {
  GeoJob &J = jobs[blockIdx.x];
  UVOL_SERIAL_PRIO();
  UVOL_DYN_SMEM(uint32_t, lds);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool ok = J.status == 0;
  const uint32_t fw = ((uint32_t)J.nf + 31) / 32, vw = (J.nverts_t[0] + 31) / 32, vcw = (uint32_t)vcap_words;
  const bool v_in_lds = vw <= vcw;
  const uint32_t stg_off = (fw + vcw + 3u) & ~3u;          // WALK_STG_DWORDS of output staging + WALK_PUB_DWORDS behind the bitmaps
  if (ok) for (uint32_t k = tid; k < fw + vcw; k += 128) lds[k] = 0;
  if (tid < WALK_PUB_DWORDS) lds[stg_off + WALK_STG_DWORDS + tid] = 0;
  __syncthreads();
  if (!ok) return;
  UVOL_L(uint32_t) stg = UVOL_TO_L(uint32_t, lds) + stg_off; UVOL_L(uint32_t) pub = stg + WALK_STG_DWORDS;
  if (v_in_lds) {
    if (wave == 1) { if (pf) walk_prefetch_wave(J.rec[0], (uint32_t)((R8 ? 32 : 64) * (size_t)J.nf), pub, R8 ? 4 : 3); return; }
    eb_walk_coop<R8>(J, UVOL_TO_L(uint32_t, lds), fw, pub, pf);
    pub[1] = 1u;
    return;
  }
  if (tid != 0) return;
  eb_walk_lane0<R8>(J, UVOL_TO_L(uint32_t, lds), UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.vvis)), stg);
}
