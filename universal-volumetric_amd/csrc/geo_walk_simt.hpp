// geo_walk_simt.hpp - K4 / K5: lane-per-walker forms of the walk and the traversals.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
// ------------------------------------------------------------------------------------------------
// Lane-per-walker forms of K4 / K5.  The wave-per-walker kernels above keep one dependent-load chain per wave and their
// visited bitmaps in LDS, which caps a CU at 3 - 6 walkers.  Here every LANE walks its own frame (or table of a frame): plain
// SIMT code, divergent branches, nothing in LDS, so the number of chains in flight is bounded by frames in HBM, not by LDS.
// The face-visited flag lives in the 4th slot of the face's own record block (it arrives with the record prefetch: no face
// bitmap), the vertex-visited bitmap is a per-walker word array in global memory touched by its one lane only (plain
// load / OR / store: a thread always sees its own stores).  `W` = lanes used per wave: few walkers are spread over many
// waves (less branch serialisation per step), many walkers are packed up to 64 per wave.
// Results are identical to the wave-per-walker kernels (same traversal, same output arrays).
// ------------------------------------------------------------------------------------------------
// `rec` is a typed global pointer (UVOL_G): global_load / global_store with exactly counted waits.  Through generic pointers
// every access was a flat_* instruction followed by s_waitcnt vmcnt(0) lgkmcnt(0), i.e. each step also waited for its own
// stores and for the neighbour prefetches it had just issued.
#define S_REC(code, vi, rc, lc)                                                                                         \
  do {                                                                                                                  \
    if (R8) { const uvol_u2 q_ = *(UVOL_G(const uvol_u2))(rec + 2 * (size_t)(code)); rec8_dec(q_.x, q_.y, vi, rc, lc); }  \
    else { const uvol_i4 q_ = *(UVOL_G(const uvol_i4))(rec + 4 * (size_t)(code)); vi = q_.x; rc = q_.y; lc = q_.z; }      \
  } while (0)
#define S_FLAG(code) (rec[(R8 ? 2 : 4) * (size_t)((code) | 3)])
// raw prefetch of a neighbour's record + its face flag (decoded only by the branch that moves there)
#define S_PRE(code, a, b, c, fl)                                                                                        \
  do {                                                                                                                  \
    if (R8) { const uvol_u2 q_ = *(UVOL_G(const uvol_u2))(rec + 2 * (size_t)(code)); a = q_.x; b = q_.y; c = 0; }          \
    else { const uvol_i4 q_ = *(UVOL_G(const uvol_i4))(rec + 4 * (size_t)(code)); a = (uint32_t)q_.x; b = (uint32_t)q_.y; c = (uint32_t)q_.z; } \
    fl = S_FLAG(code);                                                                                                  \
  } while (0)
#define S_TAKE(a, b, c, vi, rc, lc) do { if (R8) rec8_dec(a, b, vi, rc, lc); else { vi = (int)(a); rc = (int)(b); lc = (int)(c); } } while (0)

// One step of a lane is the SAME straight-line code whatever its symbol (C / R / L / S differ only in predicated selects and
// two predicated stack stores), so the lanes of a wave do not serialise on their symbols: frames of a real sequence have
// different connectivity and walk different paths, and the earlier branch-per-symbol form ran 3 - 4 x slower on them than on
// the bench's lattice frames, whose walkers happen to move in lock step (tools/exp_r3e: 450 vs 136 ms per 2160 frames, equal
// with one lane per wave).  Only the rare events leave the line: a dead end (E: pop the stack, a dependent load) and the search
// for the next component.  The S symbol is "go right and push the left neighbour": its record is already prefetched.
template <bool R8>
__device__ inline void eb_walk_simt(GeoJob &J) {
  const int nf = (int)J.nf;
  UVOL_G(uint32_t) rec = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.rec[0]));
  UVOL_G(uint32_t) vbits = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.vvis));
  UVOL_G(int32_t) proc = UVOL_TO_G(int32_t, J.proc); UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.stack); UVOL_G(int32_t) initc = UVOL_TO_G(int32_t, J.initc);
  UVOL_G(uint8_t) symb = UVOL_TO_G(uint8_t, J.symb); UVOL_G(uint8_t) start_bits = UVOL_TO_G(uint8_t, J.start_bits);
  const bool rl = J.relabel != 0; UVOL_G(const int32_t) s_of_o = UVOL_TO_G(const int32_t, J.s_of_o);
  int nproc = 0, ninit = 0, nstart = 0, nsplit = 0;
  int fo = 0, sp = 0, x = -1, vi = 0, rcn = -1, lcn = -1;
  for (;;) {
    if (x < 0) {                                          // rare: a corner to go on from - the stack, else the next component
      bool finished = false;
      for (;;) {
        if (sp > 0) {
          const int c = stack[sp - 1];
          if (c < 0 || S_FLAG(c)) { sp--; continue; }
          x = c; S_REC(x, vi, rcn, lcn);
          break;
        }
        if (fo >= nf || nproc + ninit >= nf) { finished = true; break; }
        const int f0 = rl ? s_of_o[fo] : fo;             // component starts follow the ORIGINAL face order
        fo++;
        if (S_FLAG(4 * f0)) continue;
        int v0[3], r0_[3], l0_[3];
        for (int k = 0; k < 3; k++) S_REC(4 * f0 + k, v0[k], r0_[k], l0_[k]);
        const int o0[3] = { r0_[2], r0_[0], r0_[1] };                       // opposite(k) = right field of corner (k + 2) % 3
        int interior = 1, start = 4 * f0;
        for (int k = 0; k < 3; k++) {
          if (o0[k] < 0) { interior = 0; start = 4 * f0 + k; break; }
          if (v0[k] & 1) {                // boundary vertex: swing right to the boundary edge
            int ci = 4 * f0 + k, rc = ci;
            while (rc >= 0) { ci = rc; int v_, r_, l_; S_REC(rc, v_, r_, l_); rc = l_ < 0 ? -1 : code_prv(l_); }      // left field = opposite(prev): swing right
            interior = 0; start = code_prv(ci); break;
          }
        }
        start_bits[nstart] = (uint8_t)interior;
        nstart++;
        int from;
        if (interior) {
          for (int k = 0; k < 3; k++) { const int v = v0[k] >> 1; const uint32_t w = vbits[v >> 5]; vbits[v >> 5] = w | (1u << (v & 31)); }
          S_FLAG(4 * f0) = 1u;
          initc[ninit] = 3 * f0 + 1;
          ninit++;
          from = o0[1];
          if (from < 0 || S_FLAG(from)) continue;
        } else from = start;
        stack[0] = from; sp = 1;
      }
      if (finished) break;
    }
    // ---- the common step ----
    S_FLAG(x) = 1u;
    uint32_t ra, rb, rc_, rfl, la, lb, lc_, lfl;
    S_PRE(rcn < 0 ? x : rcn, ra, rb, rc_, rfl);
    S_PRE(lcn < 0 ? x : lcn, la, lb, lc_, lfl);
    proc[nproc] = 3 * (x >> 2) + (x & 3);
    const int v = vi >> 1;
    const uint32_t vw = vbits[v >> 5];
    vbits[v >> 5] = vw | (1u << (v & 31));                                  // (already set when the tip was visited)
    const uint32_t vvis = (vw >> (v & 31)) & 1u;
    const uint32_t rvis = (rcn < 0 || rfl != 0) ? 1u : 0u, lvis = (lcn < 0 || lfl != 0) ? 1u : 0u;
    const bool ccase = ((vvis | (uint32_t)vi) & 1u) == 0;                    // tip unvisited and not on a boundary
    const uint32_t sym = ccase ? 0u : 1u + 2u * lvis + 4u * rvis;           // C 0, S 1, L 3, R 5, E 7
    symb[nproc] = (uint8_t)sym;
    nproc++;
    if (sym == 1u) { stack[sp - 1] = lcn; stack[sp] = rcn; sp++; nsplit++; }   // S: the left neighbour waits on the stack, the walk goes right
    if (sym == 7u) { sp--; x = -1; }
    else {
      const bool go_l = sym == 5u;
      x = go_l ? lcn : rcn;
      const uint32_t qa = go_l ? la : ra, qb = go_l ? lb : rb, qc = go_l ? lc_ : rc_;
      S_TAKE(qa, qb, qc, vi, rcn, lcn);
    }
  }
  J.nsym = nproc; J.nsplit = nsplit; J.nstart = nstart; J.ninit = ninit;
  if (nproc + ninit != nf) J.status = -10;
  J.rb[0].n = (uint32_t)nstart;
  uint32_t z = 0; for (int i = 0; i < nstart; i++) z += start_bits[i] == 0;
  J.rb[0].zeros = z;
}
template <bool R8>
__global__ void __launch_bounds__(64) k_eb_walk_simt(GeoJob *jobs, int n, int W) {
  const int lane = (int)threadIdx.x;
  if (lane >= W) return;
  const int j = (int)blockIdx.x * W + lane;
  if (j >= n) return;
  GeoJob &J = jobs[j];
  if (J.status != 0) return;
  eb_walk_simt<R8>(J);
}

// attribute sequencing with the same straight-line step (see eb_walk_simt)
template <bool R8>
__device__ inline void traverse_simt(GeoJob &J, int t) {
  const int nf = (int)J.nf;
  UVOL_G(uint32_t) rec = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.rec[1 + t]));
  UVOL_G(uint32_t) vbits = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.t_vvis[t]));
  UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.t_stack[t]); UVOL_G(int32_t) order = UVOL_TO_G(int32_t, J.order[t]);
  UVOL_G(const int32_t) tstart = UVOL_TO_G(const int32_t, J.tstart);
  const bool virt = J.tstart != nullptr;
  int n = 0, nvis = 0, f = 0, sp = 0, x = -1, vi = 0, rc = -1, lc = -1;
  for (;;) {
    if (x < 0) {                                          // rare: the stack, else the next unvisited face starts a component
      bool finished = false;
      for (;;) {
        if (sp > 0) {
          const int c = stack[sp - 1];
          if (c < 0 || S_FLAG(c)) { sp--; continue; }
          x = c; S_REC(x, vi, rc, lc);
          break;
        }
        if (f >= nf || nvis >= nf) { finished = true; break; }
        const int x0 = virt ? tstart[f] : 4 * f;           // components start in DECODER order (tstart: encode side, stored tables)
        f++;
        if (S_FLAG(x0)) continue;
        stack[0] = x0; sp = 1;
        const int xn = code_nxt(x0), xp = code_prv(x0);
        int vn, vp, r_, l_; S_REC(xn, vn, r_, l_); S_REC(xp, vp, r_, l_); vn >>= 1; vp >>= 1;
        uint32_t w = vbits[vn >> 5];
        if (!((w >> (vn & 31)) & 1u)) { vbits[vn >> 5] = w | (1u << (vn & 31)); order[n] = corner_of_code(xn); n++; }
        w = vbits[vp >> 5];
        if (!((w >> (vp & 31)) & 1u)) { vbits[vp >> 5] = w | (1u << (vp & 31)); order[n] = corner_of_code(xp); n++; }
      }
      if (finished) break;
    }
    S_FLAG(x) = 1u;
    nvis++;
    uint32_t ra, rb, rc_, rfl, la, lb, lc_, lfl;
    S_PRE(rc < 0 ? x : rc, ra, rb, rc_, rfl);
    S_PRE(lc < 0 ? x : lc, la, lb, lc_, lfl);
    const int v = vi >> 1;
    const uint32_t vw = vbits[v >> 5];
    vbits[v >> 5] = vw | (1u << (v & 31));
    const uint32_t vvis = (vw >> (v & 31)) & 1u;
    if (!vvis) { order[n] = 3 * (x >> 2) + (x & 3); n++; }                  // a vertex seen for the first time takes the next place
    const uint32_t rvis = (rc < 0 || rfl != 0) ? 1u : 0u, lvis = (lc < 0 || lfl != 0) ? 1u : 0u;
    const bool ccase = ((vvis | (uint32_t)vi) & 1u) == 0;
    const uint32_t k = ccase ? 0u : 1u + rvis + 2u * lvis;                  // 0 / 3: right; 2: left; 1: fork (right, left waits); 4: dead end
    if (k == 1u) { stack[sp - 1] = lc; stack[sp] = rc; sp++; }
    if (k == 4u) { sp--; x = -1; }
    else {
      const bool go_l = k == 2u;
      x = go_l ? lc : rc;
      const uint32_t qa = go_l ? la : ra, qb = go_l ? lb : rb, qc = go_l ? lc_ : rc_;
      S_TAKE(qa, qb, qc, vi, rc, lc);
    }
  }
  J.ne[t] = (uint32_t)n;
  if (t == 0 && J.nverts != 0xffffffffu && (uint32_t)n != J.nverts) J.status = -11;      // (the decode path has no expected count)
}
// walker id = table * n + frame: the lanes of a wave walk the same table of consecutive frames (similar lengths)
template <bool R8>
__global__ void __launch_bounds__(64) k_traverse_simt(GeoJob *jobs, int n, int W) {
  const int lane = (int)threadIdx.x;
  if (lane >= W) return;
  const int id = (int)blockIdx.x * W + lane;
  if (id >= 3 * n) return;
  const int t = id / n, j = id - t * n;                  // (the three tables of ONE frame in neighbouring lanes was measured slower: 347 vs 307 / 202 ms)
  GeoJob &J = jobs[j];
  const int ai = t > 0 ? t - 1 : 0;
  if (J.status != 0 || (t > 0 && (ai >= J.nad || !J.interior_seams[ai]))) return;
  traverse_simt<R8>(J, t);
}
#undef S_REC
#undef S_FLAG
#undef S_PRE
#undef S_TAKE

// ------------------------------------------------------------------------------------------------
// The same two lane-per-walker kernels on ONE 16-byte record per FACE (format 2, pack_face_records): the three vertex fields, the
// three opposite-corner codes and the face-visited flags.  A step still moves by corner codes (4 * face + k): the vertex of corner k
// is vertex field k, its right / left neighbours are opposite fields (k + 1) % 3 / (k + 2) % 3 of the same record.  Against the
// 8-byte corner records this halves the bytes the walkers fetch and write back (the flag dirties the line it is in), puts eight faces
// instead of four on a 128-byte line (more of a walker's dependent loads hit a line a neighbouring face already brought in), brings a
// candidate's record AND its visited flag in one load instead of two, and halves the record tables.
// Two flags per record: bit 63 (dword 1) is the edgebreaker walk's, bit 127 (dword 3) the base-table traversal's - the base table of
// the traversals is the walk's own table (round 5: the decoder's face order is virtual, GeoJob::tstart), so it is never packed twice.
// The attribute tables with seams have tables of their own and use bit 63.  FD = the dword that holds this walker's flag.
// Batches whose face count or id space does not fit the 21-bit fields keep the 16-byte corner records (geo_rec8).
// (Measured in round 4 and removed: the flags as one bit per face in an array of their own - clean record lines, two more loads per
// step, 275 against 251 ms - and loads that bypass the L1, 234 against 206 ms; docs/HISTORY.md.)
// ------------------------------------------------------------------------------------------------
#ifdef HIPEMU
struct uvol_u4 { uint32_t x, y, z, w; };
#else
typedef uint32_t uvol_u4 __attribute__((ext_vector_type(4)));
#endif
__device__ __forceinline__ uvol_u4 f16_load(UVOL_G(uint32_t) rec, int face) { return *(UVOL_G(const uvol_u4))(rec + 4 * (size_t)face); }
__device__ __forceinline__ void f16_dec(const uvol_u4 &q, int k, int &vi, int &rc, int &lc) {
  const uint64_t lo = (uint64_t)q.x | ((uint64_t)q.y << 32), hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
  const int s = 21 * k, sr = k == 2 ? 0 : s + 21, sl = k == 0 ? 42 : s - 21;
  vi = (int)((uint32_t)(lo >> s) & 0x1fffffu);
  rc = (int)((uint32_t)(hi >> sr) << 11) >> 11;            // 21-bit field, all ones = none
  lc = (int)((uint32_t)(hi >> sl) << 11) >> 11;
}
template <int FD> __device__ __forceinline__ bool f16_seen(const uvol_u4 &q) { return ((FD == 1 ? q.y : q.w) >> 31) != 0; }
template <int FD> __device__ __forceinline__ void f16_mark(UVOL_G(uint32_t) rec, int f, const uvol_u4 &q) { rec[4 * (size_t)f + FD] = (FD == 1 ? q.y : q.w) | 0x80000000u; }
__device__ inline void eb_walk_simt_f16(GeoJob &J) {
  const int nf = (int)J.nf;
  UVOL_G(uint32_t) rec = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.rec[0]));
  UVOL_G(uint32_t) vbits = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.vvis));
  UVOL_G(int32_t) proc = UVOL_TO_G(int32_t, J.proc); UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.stack); UVOL_G(int32_t) initc = UVOL_TO_G(int32_t, J.initc);
  UVOL_G(uint8_t) symb = UVOL_TO_G(uint8_t, J.symb); UVOL_G(uint8_t) start_bits = UVOL_TO_G(uint8_t, J.start_bits);
  const bool rl = J.relabel != 0; UVOL_G(const int32_t) s_of_o = UVOL_TO_G(const int32_t, J.s_of_o);
  int nproc = 0, ninit = 0, nstart = 0, nsplit = 0;
  int fo = 0, sp = 0, x = -1, vi = 0, rcn = -1, lcn = -1;
  uvol_u4 q; q.x = q.y = q.z = q.w = 0;
  for (;;) {
    if (x < 0) {                                          // rare: a corner to go on from - the stack, else the next component
      bool finished = false;
      for (;;) {
        if (sp > 0) {
          const int c = stack[sp - 1];
          if (c < 0) { sp--; continue; }
          const uvol_u4 qq = f16_load(rec, c >> 2);
          if (f16_seen<1>(qq)) { sp--; continue; }
          x = c; q = qq; f16_dec(q, x & 3, vi, rcn, lcn);
          break;
        }
        if (fo >= nf || nproc + ninit >= nf) { finished = true; break; }
        const int f0 = rl ? s_of_o[fo] : fo;             // component starts follow the ORIGINAL face order
        fo++;
        const uvol_u4 q0 = f16_load(rec, f0);
        if (f16_seen<1>(q0)) continue;
        int v0[3], r0_[3], l0_[3];
        for (int k = 0; k < 3; k++) f16_dec(q0, k, v0[k], r0_[k], l0_[k]);
        const int o0[3] = { r0_[2], r0_[0], r0_[1] };                       // opposite(k) = right field of corner (k + 2) % 3
        int interior = 1, start = 4 * f0;
        for (int k = 0; k < 3; k++) {
          if (o0[k] < 0) { interior = 0; start = 4 * f0 + k; break; }
          if (v0[k] & 1) {                // boundary vertex: swing right to the boundary edge
            int ci = 4 * f0 + k, rc = ci;
            while (rc >= 0) { ci = rc; int v_, r_, l_; f16_dec(f16_load(rec, rc >> 2), rc & 3, v_, r_, l_); rc = l_ < 0 ? -1 : code_prv(l_); }      // left field = opposite(prev): swing right
            interior = 0; start = code_prv(ci); break;
          }
        }
        start_bits[nstart] = (uint8_t)interior;
        nstart++;
        int from;
        if (interior) {
          for (int k = 0; k < 3; k++) { const int v = v0[k] >> 1; const uint32_t w = vbits[v >> 5]; vbits[v >> 5] = w | (1u << (v & 31)); }
          f16_mark<1>(rec, f0, q0);
          initc[ninit] = 3 * f0 + 1;
          ninit++;
          from = o0[1];
          if (from < 0) continue;
          if (f16_seen<1>(f16_load(rec, from >> 2))) continue;
        } else from = start;
        stack[0] = from; sp = 1;
      }
      if (finished) break;
    }
    // ---- the common step (straight-line, see eb_walk_simt) ----
    const int f = x >> 2, rf = (rcn < 0 ? x : rcn) >> 2, lf = (lcn < 0 ? x : lcn) >> 2;
    f16_mark<1>(rec, f, q);
    const uvol_u4 qr = f16_load(rec, rf), ql = f16_load(rec, lf);
    proc[nproc] = 3 * f + (x & 3);
    const int v = vi >> 1;
    const uint32_t vw = vbits[v >> 5];
    vbits[v >> 5] = vw | (1u << (v & 31));                                  // (already set when the tip was visited)
    const uint32_t vvis = (vw >> (v & 31)) & 1u;
    const uint32_t rvis = (rcn < 0 || f16_seen<1>(qr)) ? 1u : 0u, lvis = (lcn < 0 || f16_seen<1>(ql)) ? 1u : 0u;
    const bool ccase = ((vvis | (uint32_t)vi) & 1u) == 0;                    // tip unvisited and not on a boundary
    const uint32_t sym = ccase ? 0u : 1u + 2u * lvis + 4u * rvis;           // C 0, S 1, L 3, R 5, E 7
    symb[nproc] = (uint8_t)sym;
    nproc++;
    if (sym == 1u) { stack[sp - 1] = lcn; stack[sp] = rcn; sp++; nsplit++; if (sp > (int)J.stcap) { J.status = GEO_E_WS_OVERFLOW; break; } }   // S: the left neighbour waits on the stack, the walk goes right
    if (sym == 7u) { sp--; x = -1; }
    else {
      const bool go_l = sym == 5u;
      x = go_l ? lcn : rcn;
      q.x = go_l ? ql.x : qr.x; q.y = go_l ? ql.y : qr.y; q.z = go_l ? ql.z : qr.z; q.w = go_l ? ql.w : qr.w;
      f16_dec(q, x & 3, vi, rcn, lcn);
    }
  }
  if (J.status != 0) return;
  J.nsym = nproc; J.nsplit = nsplit; J.nstart = nstart; J.ninit = ninit;
  if (nproc + ninit != nf) J.status = -10;
  J.rb[0].n = (uint32_t)nstart;
  uint32_t z = 0; for (int i = 0; i < nstart; i++) z += start_bits[i] == 0;
  J.rb[0].zeros = z;
}
__global__ void __launch_bounds__(64) k_eb_walk_simt_f16(GeoJob *jobs, int n, int W) {
  const int lane = (int)threadIdx.x;
  if (lane >= W) return;
  const int j = (int)blockIdx.x * W + lane;
  if (j >= n) return;
  GeoJob &J = jobs[j];
  if (J.status != 0) return;
  eb_walk_simt_f16(J);
}
// Depth-first sequencing of table t.  Components start at the first unvisited face IN DECODER ORDER, at the decoder's corner 0 of that
// face: face f itself on the decode path (tstart == nullptr), tstart[f] on the encode side, whose tables keep the stored order.
template <int FD>
__device__ inline void traverse_simt_f16(GeoJob &J, int t) {
  const int nf = (int)J.nf;
  UVOL_G(uint32_t) rec = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.rec[1 + t]));
  UVOL_G(uint32_t) vbits = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.t_vvis[t]));
  UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.t_stack[t]); UVOL_G(int32_t) order = UVOL_TO_G(int32_t, J.order[t]);
  UVOL_G(const int32_t) tstart = UVOL_TO_G(const int32_t, J.tstart);
  const bool virt = J.tstart != nullptr;
  int n = 0, nvis = 0, f = 0, sp = 0, x = -1, vi = 0, rc = -1, lc = -1;
  uvol_u4 q; q.x = q.y = q.z = q.w = 0;
  for (;;) {
    if (x < 0) {                                          // rare: the stack, else the next unvisited face starts a component
      bool finished = false;
      for (;;) {
        if (sp > 0) {
          const int c = stack[sp - 1];
          if (c < 0) { sp--; continue; }
          const uvol_u4 qq = f16_load(rec, c >> 2);
          if (f16_seen<FD>(qq)) { sp--; continue; }
          x = c; q = qq; f16_dec(q, x & 3, vi, rc, lc);
          break;
        }
        if (f >= nf || nvis >= nf) { finished = true; break; }
        // the next unvisited face in decoder order.  A table with seams falls into many components (one per chart), and between two
        // of them this scan passes every face once: eight flag words per round trip instead of one record (a lane that scans holds up
        // the other walkers of its wave, and a walker alone spent a sixth of its time here)
        int hit = -1, x0 = 0;
        while (f < nf) {
          int xs[8]; uint32_t y[8];
          for (int k = 0; k < 8; k++) { const int fk = f + k < nf ? f + k : nf - 1; xs[k] = virt ? tstart[fk] : 4 * fk; }
          for (int k = 0; k < 8; k++) y[k] = rec[4 * (size_t)(xs[k] >> 2) + FD];
          uint32_t m = 0;
          for (int k = 0; k < 8; k++) m |= ((y[k] >> 31) ^ 1u) << k;
          if (nf - f < 8) m &= (1u << (nf - f)) - 1u;
          if (m) { const int k0 = __builtin_ctz(m); hit = f + k0; x0 = xs[0]; for (int k = 1; k < 8; k++) x0 = k == k0 ? xs[k] : x0; break; }
          f += 8;
        }
        if (hit < 0) { finished = true; break; }
        f = hit + 1;
        const int f0 = x0 >> 2;
        const uvol_u4 q0 = f16_load(rec, f0);
        if (f16_seen<FD>(q0)) continue;
        stack[0] = x0; sp = 1;
        const int xn = code_nxt(x0), xp = code_prv(x0);
        int vn, vp, r_, l_; f16_dec(q0, xn & 3, vn, r_, l_); f16_dec(q0, xp & 3, vp, r_, l_); vn >>= 1; vp >>= 1;
        uint32_t w = vbits[vn >> 5];
        if (!((w >> (vn & 31)) & 1u)) { vbits[vn >> 5] = w | (1u << (vn & 31)); order[n] = corner_of_code(xn); n++; }
        w = vbits[vp >> 5];
        if (!((w >> (vp & 31)) & 1u)) { vbits[vp >> 5] = w | (1u << (vp & 31)); order[n] = corner_of_code(xp); n++; }
      }
      if (finished) break;
    }
    const int fc = x >> 2, rf = (rc < 0 ? x : rc) >> 2, lf = (lc < 0 ? x : lc) >> 2;
    f16_mark<FD>(rec, fc, q);
    nvis++;
    const uvol_u4 qr = f16_load(rec, rf), ql = f16_load(rec, lf);
    const int v = vi >> 1;
    const uint32_t vw = vbits[v >> 5];
    vbits[v >> 5] = vw | (1u << (v & 31));
    const uint32_t vvis = (vw >> (v & 31)) & 1u;
    if (!vvis) { order[n] = 3 * fc + (x & 3); n++; }                        // a vertex seen for the first time takes the next place
    const uint32_t rvis = (rc < 0 || f16_seen<FD>(qr)) ? 1u : 0u, lvis = (lc < 0 || f16_seen<FD>(ql)) ? 1u : 0u;
    const bool ccase = ((vvis | (uint32_t)vi) & 1u) == 0;
    const uint32_t k = ccase ? 0u : 1u + rvis + 2u * lvis;                  // 0 / 3: right; 2: left; 1: fork (right, left waits); 4: dead end
    if (k == 1u) { stack[sp - 1] = lc; stack[sp] = rc; sp++; if (J.stcap && sp > (int)J.stcap) { J.status = GEO_E_WS_OVERFLOW; break; } }      // (stcap 0: decode path, a slot per face)
    if (k == 4u) { sp--; x = -1; }
    else {
      const bool go_l = k == 2u;
      x = go_l ? lc : rc;
      q.x = go_l ? ql.x : qr.x; q.y = go_l ? ql.y : qr.y; q.z = go_l ? ql.z : qr.z; q.w = go_l ? ql.w : qr.w;
      f16_dec(q, x & 3, vi, rc, lc);
    }
  }
  if (J.status != 0) return;
  J.ne[t] = (uint32_t)n;
  if (t == 0 && J.nverts != 0xffffffffu && (uint32_t)n != J.nverts) J.status = -11;      // (the decode path has no expected count)
}
// walker id = table * n + frame (tables t0 .. t0 + nt - 1): the lanes of a wave walk the same table of consecutive frames
__global__ void __launch_bounds__(64) k_traverse_simt_f16(GeoJob *jobs, int n, int W, int t0, int nt) {
  const int lane = (int)threadIdx.x;
  if (lane >= W) return;
  const int id = (int)blockIdx.x * W + lane;
  if (id >= nt * n) return;
  const int t = t0 + id / n, j = id % n;
  GeoJob &J = jobs[j];
  const int ai = t > 0 ? t - 1 : 0;
  if (J.status != 0 || (t > 0 && (ai >= J.nad || !J.interior_seams[ai]))) return;
  if (t == 0 && J.base_hi) traverse_simt_f16<3>(J, t); else traverse_simt_f16<1>(J, t);
}

// ------------------------------------------------------------------------------------------------
// Wave form of the lane-per-walker traverser on per-face records (round 5).  W walkers share a wave (lanes 0 .. W - 1) and ALL 64 lanes
// stay in the loop, because what kept several walkers from sharing a wave was not the common step but the rare ones (round 4: 293 / 372 /
// 471 ms per 2560 frames with 1 / 2 / 4 walkers per wave): every lane waited while ONE lane popped its stack (two dependent round trips)
// or searched the next component (a round trip per eight faces) - and a one-lane wave still holds its SIMD's vector pipe for four cycles
// per instruction, so 3840 one-walker waves took half of the chip's vector issue slots away from the kernels running beside them.  Here
//  * a POP is an ordinary step: the walker stands on a dummy face (index nf: visited, no neighbours) whose right neighbour is the top of
//    the pending stack and whose tip is a dummy vertex that counts as visited.  The step then says "go right" (the candidate is unvisited:
//    the walk continues there) or "dead end" (it was visited meanwhile: the next pending corner is tried) - the same instructions as a real
//    step, no branch.  The top of the stack is carried in a register and its successor is fetched with the step's other loads.
//  * "no neighbour" is the dummy face as well (min(code >> 2, nf) instead of a compare and a select per side); the vertex order entry and
//    the pending corner are stored unconditionally at their cursors and only the cursors are predicated.
//  * the search for the next component is done by the WHOLE wave for one walker at a time: 64 faces per round trip (one per lane, first
//    hit by a ballot) instead of eight, through the walker's pointers broadcast with v_readlane.
// Same order[] as every other form.  Pending stack here = the left neighbours of the forks only (the other forms also keep the current
// path's placeholder on it).
// ------------------------------------------------------------------------------------------------
#ifdef HIPEMU
#define UVOL_BCAST64(v, l) ((uint64_t)UVOL_READLANE((uint32_t)(v), l) | ((uint64_t)UVOL_READLANE((uint32_t)((uint64_t)(v) >> 32), l) << 32))
#else
#define UVOL_BCAST64(v, l) ((uint64_t)UVOL_READLANE((uint32_t)(v), l) | ((uint64_t)UVOL_READLANE((uint32_t)((uint64_t)(v) >> 32), l) << 32))
#endif
template <int FD>
__device__ __forceinline__ void traverse_wave_f16(GeoJob *jobs, int n, int W, int t) {
  const int lane = (int)threadIdx.x;
  const int j = (int)blockIdx.x * W + lane;
  const int ai = t > 0 ? t - 1 : 0;
  bool live = lane < W && j < n;
  if (live) { const GeoJob &J0 = jobs[j]; live = J0.status == 0 && !(t > 0 && (ai >= J0.nad || !J0.interior_seams[ai])); }
  GeoJob &J = jobs[live ? j : 0];
  const int nf = live ? (int)J.nf : 0;
  UVOL_G(uint32_t) rec = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.rec[1 + t]));
  UVOL_G(uint32_t) vbits = UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.t_vvis[t]));
  UVOL_G(int32_t) pend_s = UVOL_TO_G(int32_t, J.t_stack[t]); UVOL_G(int32_t) order = UVOL_TO_G(int32_t, J.order[t]);
  UVOL_G(const int32_t) tstart = UVOL_TO_G(const int32_t, J.tstart);
  const uint32_t stcap = live ? (J.stcap ? J.stcap : 0x7fffffffu) : 0u;      // (0: the decode path, whose stacks hold a corner per face)
  const uint32_t vdum = live ? J.nverts_t[1 + t] : 0u;     // dummy vertex: the bit past every id of the table (the bitmaps have 512 bits of slack)
  if (live) vbits[vdum >> 5] = vbits[vdum >> 5] | (1u << (vdum & 31));
  int n_ord = 0, nvis = 0, f = 0, pend = 0, top = 0, x = -1, vi = 0, rc = -1, lc = -1;
  bool done = !live;
  uvol_u4 q; q.x = q.z = 0; q.y = q.w = 0x80000000u;
  for (;;) {
    // ---- walkers without a face to stand on and nothing pending: the next component, searched by the whole wave, one walker at a time ----
    unsigned long long need = __ballot(!done && x < 0 && pend == 0);
    while (need) {
      const int l = (int)__ffsll((long long)need) - 1; need &= need - 1;
      const uint64_t rec_u = UVOL_BCAST64((uint64_t)(uintptr_t)rec, l), ts_u = UVOL_BCAST64((uint64_t)(uintptr_t)tstart, l);
      const int nf_u = (int)UVOL_READLANE(nf, l), nvis_u = (int)UVOL_READLANE(nvis, l);
      int f_u = (int)UVOL_READLANE(f, l), x0 = -1;
      UVOL_G(uint32_t) rec_l = (UVOL_G(uint32_t))(uintptr_t)rec_u; UVOL_G(const int32_t) ts_l = (UVOL_G(const int32_t))(uintptr_t)ts_u;
      if (nvis_u < nf_u) {
        while (f_u < nf_u) {
          const int fk = f_u + lane;
          int xs = -1; uint32_t y = 0x80000000u;
          if (fk < nf_u) { xs = ts_u ? ts_l[fk] : 4 * fk; y = rec_l[4 * (size_t)(xs >> 2) + FD]; }
          const unsigned long long m = __ballot((y >> 31) == 0);
          if (m) { const int k0 = (int)__ffsll((long long)m) - 1; x0 = (int)UVOL_READLANE(xs, k0); f_u += k0 + 1; break; }
          f_u += 64;
        }
      }
      if (lane == l) {
        f = f_u;
        if (x0 < 0) done = true;
        else {                                            // the component starts at the decoder's corner 0 of that face; its two other vertices come first
          const uvol_u4 q0 = f16_load(rec, x0 >> 2);
          const int xn = code_nxt(x0), xp = code_prv(x0);
          int vn, vp, r_, l_; f16_dec(q0, xn & 3, vn, r_, l_); f16_dec(q0, xp & 3, vp, r_, l_); vn >>= 1; vp >>= 1;
          uint32_t w = vbits[vn >> 5];
          if (!((w >> (vn & 31)) & 1u)) { vbits[vn >> 5] = w | (1u << (vn & 31)); order[n_ord] = corner_of_code(xn); n_ord++; }
          w = vbits[vp >> 5];
          if (!((w >> (vp & 31)) & 1u)) { vbits[vp >> 5] = w | (1u << (vp & 31)); order[n_ord] = corner_of_code(xp); n_ord++; }
          x = x0; q = q0; f16_dec(q, x & 3, vi, rc, lc);
        }
      }
    }
    if (!__ballot(!done)) break;
    if (!done) {
      // ---- one step: on face x, or (x < 0) on the dummy face with the top of the pending stack to its right ----
      const bool run = x >= 0;
      const int fc = run ? x >> 2 : nf;
      const int rcc = run ? rc : top, lcc = run ? lc : -1;
      const uint32_t vic = run ? (uint32_t)vi : ((vdum << 1) | 1u);
      const uint32_t unf = (uint32_t)nf;
      const uint32_t rf = ((uint32_t)rcc >> 2) < unf ? ((uint32_t)rcc >> 2) : unf, lf = ((uint32_t)lcc >> 2) < unf ? ((uint32_t)lcc >> 2) : unf;
      f16_mark<FD>(rec, fc, q);
      const uvol_u4 qr = f16_load(rec, (int)rf), ql = f16_load(rec, (int)lf);
      const uint32_t v = vic >> 1;
      const uint32_t vw = vbits[v >> 5];
      const int top2 = pend_s[pend >= 2 ? pend - 2 : 0];                    // the pending corner below the top (a pop needs it next)
      vbits[v >> 5] = vw | (1u << (v & 31));
      const uint32_t vvis = (vw >> (v & 31)) & 1u;
      order[n_ord] = 3 * fc + (x & 3);                                        // a vertex seen for the first time takes the next place
      n_ord += (int)(vvis ^ 1u);
      nvis += run ? 1 : 0;
      const uint32_t rvis = f16_seen<FD>(qr) ? 1u : 0u, lvis = f16_seen<FD>(ql) ? 1u : 0u;
      const bool ccase = ((vvis | vic) & 1u) == 0;
      const uint32_t k = ccase ? 0u : 1u + rvis + 2u * lvis;                  // 0 / 3: right; 2: left; 1: fork (right, left waits); 4: dead end
      pend_s[pend] = lcc;                                                     // (kept only by a fork)
      const bool fork = k == 1u;
      if (!run) { pend--; top = top2; }                                       // the candidate is used up either way
      if (fork) { pend++; top = lcc; if (pend > (int)stcap) { J.status = GEO_E_WS_OVERFLOW; done = true; } }
      if (k == 4u) { x = -1; q.x = q.z = 0; q.y = q.w = 0x80000000u; }
      else {
        const bool go_l = k == 2u;
        x = go_l ? lcc : rcc;
        q.x = go_l ? ql.x : qr.x; q.y = go_l ? ql.y : qr.y; q.z = go_l ? ql.z : qr.z; q.w = go_l ? ql.w : qr.w;
        f16_dec(q, x & 3, vi, rc, lc);
      }
    }
  }
  if (live && J.status == 0) {
    J.ne[t] = (uint32_t)n_ord;
    if (t == 0 && J.nverts != 0xffffffffu && (uint32_t)n_ord != J.nverts) J.status = -11;      // (the decode path has no expected count)
  }
}
// grid (waves, tables): table t0 + blockIdx.y, walkers blockIdx.x * W .. of that table in lanes 0 .. W - 1
__global__ void __launch_bounds__(64) k_traverse_wave_f16(GeoJob *jobs, int n, int W, int t0, int base_hi) {
  const int t = t0 + (int)blockIdx.y;
  if (t == 0 && base_hi) traverse_wave_f16<3>(jobs, n, W, t); else traverse_wave_f16<1>(jobs, n, W, t);
}
// (Measured in round 5 and removed, docs/HISTORY.md: a cooperative form of both walkers on per-face records - one walker per wave, lanes 0 / 1
// fetch the neighbours' records, lane 2 the vertex word, state in SGPRs like the LDS walkers of small batches - with the flags in the
// records and the vertex bits in global memory: 0.69 us per face at 320 frames, 1.3 us at 2560 (one scalar unit per CU), against 0.32 us
// for the LDS walk: what makes the LDS walkers fast is that nothing they test per step has to come from L2, not the division of labour.)
