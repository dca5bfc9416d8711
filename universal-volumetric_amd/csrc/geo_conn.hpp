// geo_conn.hpp - K4b: inverse maps, split events, valence replay, context streams, seams, attribute vertices.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
// face_time[f] = index of the symbol that encoded stored face f; the faces that only start a component (interior start faces, no
// symbol) get -(j + 2), j = their index in initc[].  The inverse of proc[], built in parallel so that the serial walker has no scatter
// store in its loop.  The same pass writes the DECODER'S face order (decoder face f <-> processed corner nsym - 1 - f,
// then the interior start faces; decoder corner 3f + k <-> rot^k of that corner) as tstart[f] = code of the stored corner that is the
// decoder's corner 3f: the only form in which the decoder's numbering exists on the encode side (see GeoJob::tstart).
__global__ void __launch_bounds__(UVOL_BLOCK) k_face_time(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (i < (uint32_t)J.nsym) { const int c = J.proc[i]; J.face_time[c / 3] = (int32_t)i; J.tstart[J.nsym - 1 - (int)i] = code_of_corner(c); }
  if (i < (uint32_t)J.ninit) { const int c = J.initc[i]; J.face_time[c / 3] = -(int32_t)i - 2; J.tstart[J.nsym + (int)i] = code_of_corner(c); }
}
__device__ __forceinline__ int att_vertex(const GeoJob &J, int slot, int c);
// the vertex field of corner c in the record table of traversal table t (any format)
__device__ __forceinline__ int rec_vertex_field(const GeoJob &J, int t, int c, int r8) {
  if (r8 == 2) { const uint32_t *q = reinterpret_cast<const uint32_t *>(J.rec[1 + t]) + 4 * (size_t)(c / 3); return (int)((uint32_t)((((uint64_t)q[1] << 32) | q[0]) >> (21 * (c % 3))) & 0x1fffffu); }
  const size_t code = (size_t)code_of_corner(c);
  return r8 ? (int)((uint32_t)J.rec[1 + t][2 * code] & 0x1fffffu) : J.rec[1 + t][4 * code];
}
// v2d[t][vertex] = position of the vertex in the coding order of table t: the inverse of order[t][] (same reason).  On the encode
// side (quantised values by id present) the pass also takes the minimum / maximum of the quantised values that are CODED (the wrap
// transform's bounds run over the entries, not over every value of the input arrays): positions with table 0, texture coordinates
// with the table their attribute is sequenced by.
__global__ void __launch_bounds__(UVOL_BLOCK) k_v2d(GeoJob *jobs, int r8) {
  GeoJob &J = jobs[blockIdx.y];
  const int t = blockIdx.z;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const bool live = J.status == 0 && !(t > 0 && (t - 1 >= J.nad || !J.interior_seams[t - 1]));
  if (blockIdx.x * UVOL_BLOCK >= (live ? J.ne[t] : 0u)) return;            // block-uniform
  int lo0 = 0x7fffffff, hi0 = -0x7fffffff - 1, lo1 = 0x7fffffff, hi1 = -0x7fffffff - 1;
  int iu = -1;                                                            // the texture-coordinate slot, if this table sequences it
  if (J.qpos) for (int k = 0; k < J.nad; k++) if (J.att_kind[k] == 0 && (J.interior_seams[k] ? t == 1 + k : t == 0)) iu = k;
  if (i < J.ne[t]) {
    const int c = J.order[t][i];
    // the vertex of the entry's corner: out of the record table on the decode path; on the encode side from the stored corner table (one
    // 4-byte gather that table 0 shares with the position id, instead of a 16-byte record per entry)
    int vid, pid = 0;
    if (!J.qpos) vid = rec_vertex_field(J, t, c, r8) >> 1;
    else if (t == 0) { pid = J.cp[c]; vid = J.extra_v ? J.vert[c] : pid; }
    else vid = att_vertex(J, t - 1, c);
    J.v2d[t][vid] = (int32_t)i;
    if (J.qpos && t == 0) {
      const uint16_t *q = J.qpos + 4 * (size_t)pid;
      for (int k = 0; k < 3; k++) { const int v = q[k]; lo0 = v < lo0 ? v : lo0; hi0 = v > hi0 ? v : hi0; }
    }
    if (iu >= 0) {
      const uint16_t *q = J.quv + 2 * (size_t)J.cu[c];
      for (int k = 0; k < 2; k++) { const int v = q[k]; lo1 = v < lo1 ? v : lo1; hi1 = v > hi1 ? v : hi1; }
    }
  }
  if (!J.qpos) return;
  for (int d = 32; d >= 1; d >>= 1) {
    int a = __shfl_xor(lo0, d), b = __shfl_xor(hi0, d); lo0 = a < lo0 ? a : lo0; hi0 = b > hi0 ? b : hi0;
    a = __shfl_xor(lo1, d); b = __shfl_xor(hi1, d); lo1 = a < lo1 ? a : lo1; hi1 = b > hi1 ? b : hi1;
  }
  if ((threadIdx.x & 63) == 0) {
    if (t == 0 && lo0 <= hi0) { atomicMin(&J.wrap_lo[0], lo0); atomicMax(&J.wrap_hi[0], hi0); }
    if (iu >= 0 && lo1 <= hi1) { atomicMin(&J.wrap_lo[1], lo1); atomicMax(&J.wrap_hi[1], hi1); }
  }
}

// topology-split events (CheckAndStoreTopologySplitEvent): symbol i contributes an event for each already-encoded
// right / left neighbour whose own symbol is S.  flag[i] = number of events (0..2), compacted in symbol order.
__device__ inline int eb_events_of(const GeoJob &J, uint32_t i, int ev_spl[2], int ev_edge[2]) {
  const int sym = J.symb[i]; int n = 0;
  if (sym != 5 && sym != 3 && sym != 7) return 0;
  const int c = J.proc[i];
  const int rcn = J.opp[g_nxt(c)], lcn = J.opp[g_prv(c)];
  if ((sym == 5 || sym == 7) && rcn >= 0) { const int t = J.face_time[rcn / 3]; if (t >= 0 && J.symb[t] == 1) { ev_spl[n] = t; ev_edge[n] = 1; n++; } }
  if ((sym == 3 || sym == 7) && lcn >= 0) { const int t = J.face_time[lcn / 3]; if (t >= 0 && J.symb[t] == 1) { ev_spl[n] = t; ev_edge[n] = 0; n++; } }
  return n;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_eb_event_flags(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  uint32_t n = 0;
  if (i < J.nf) { int a[2], b[2]; n = i < (uint32_t)J.nsym ? (uint32_t)eb_events_of(J, i, a, b) : 0u; J.evcnt[i] = (uint8_t)n; }
  const uint32_t tot = block_sum(n);
  if (threadIdx.x == 0 && blockIdx.x < uvol_blocks_dev(J.nf)) J.bsum2[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_eb_event_compact(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  // events are rare (two per S symbol): a block whose scanned sums say 'none' has nothing to place (uniform: read by every thread
  // from the same two words)
  if (J.status == 0 && blockIdx.x != 0 && blockIdx.x < uvol_blocks_dev(J.nf) && J.bsum2[blockIdx.x + 1] == J.bsum2[blockIdx.x]) return;
  const bool live = J.status == 0 && i < J.nf;
  uint32_t v = live ? J.evcnt[i] : 0, tot;
  const uint32_t pos = block_excl_scan(v, &tot) + ((J.status == 0 && blockIdx.x <= uvol_blocks_dev(J.nf)) ? J.bsum2[blockIdx.x] : 0);
  if (live && v) {
    int a[2], b[2]; const int n = eb_events_of(J, i, a, b);
    if (pos + (uint32_t)n > J.evcap) J.status = GEO_E_WS_OVERFLOW;        // (sized for the usual handful of events, not for two per face)
    else for (int k = 0; k < n; k++) { J.ev_src[pos + k] = (int)i; J.ev_spl[pos + k] = a[k]; J.ev_edge[pos + k] = (uint8_t)b[k]; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) J.nev = (int)J.bsum2[uvol_blocks_dev(J.nf)];
}

// working copies the replay mutates: corner -> vertex map (S symbols re-map corners to new vertices) and the valence per vertex
__global__ void __launch_bounds__(UVOL_BLOCK) k_valence_init(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t t = blockIdx.x * UVOL_BLOCK + threadIdx.x, stride = gridDim.x * UVOL_BLOCK;
  { const uint32_t n4 = J.nc / 4;                                                     // 16 bytes per lane (both arrays are 16-byte aligned)
    const uint4 *src = reinterpret_cast<const uint4 *>(geo_vt(J)); uint4 *dst = reinterpret_cast<uint4 *>(J.c2vm);
    for (uint32_t q = t; q < n4; q += stride) dst[q] = src[q];
    for (uint32_t c = 4 * n4 + t; c < J.nc; c += stride) J.c2vm[c] = geo_vt(J)[c]; }
  const uint32_t nv0 = J.nverts_t[0] < J.ecap ? J.nverts_t[0] : J.ecap;
  for (uint32_t v = t; v < nv0; v += stride) J.vval[v] = J.ring_d[v];
}
// valence bookkeeping replay: ctx_of[i] = context (0..5) under which symbol i-1 is coded (i >= 1).
// The context of symbol i is the clamped valence of the vertex at next(corner_i) just before i updates it.  Between two
// split symbols valences only receive fixed decrements (C: n-1 p-1; R: a-1 n-1 p-2; L: a-1 n-2 p-1; E: a-2 n-2 p-2), so
// a run of up to 64 symbols is resolved by the whole wave at once: lane j reads the run-start valence of its vertex
// and subtracts what lanes k < j apply to that same vertex (one pass of v_readlane broadcasts), then every lane posts
// its three decrements with atomic adds.  Only an S symbol (vertex split: ring walks + corner re-mapping) is serial.
__global__ void __launch_bounds__(64) k_eb_valence(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.x];
  UVOL_SERIAL_PRIO();
  const uint32_t lane = threadIdx.x;
  const bool ok = J.status == 0;
  const int nsym = ok ? J.nsym : 0, nc = (int)J.nc;
  const int32_t *opp = J.opp, *proc = J.proc, *ftime = J.face_time; const uint8_t *symb = J.symb;
  int32_t *vval = J.vval, *c2vm = J.c2vm;
  // initial valences / corner->vertex replica: filled by k_valence_init (parallel) before this launch
  const int nv0 = ok ? (int)J.nverts_t[0] : 0;
  (void)nc;
  int nvval = nv0;
  for (int base = 0; base < nsym; base += 64) {
    const int mi = base + (int)lane;
    // lane-parallel gather of the chunk's corners, symbols and vertex ids (valid until an S re-maps corners)
    int c_ = 0, s_ = 0, va_ = 0, vn_ = 0, vp_ = 0;
    if (mi < nsym) { c_ = proc[mi]; s_ = symb[mi]; va_ = c2vm[c_]; vn_ = c2vm[g_nxt(c_)]; vp_ = c2vm[g_prv(c_)]; }
    const int cnt = nsym - base < 64 ? nsym - base : 64;
    // decrements of this lane's symbol, packed a | n << 2 | p << 4
    const uint32_t dpk = s_ == 0 ? 0x14u : (s_ == 5 ? 0x25u : (s_ == 3 ? 0x19u : 0x2au));
    int start = 0;
    while (start < cnt) {
      const unsigned long long smask = __ballot((int)lane >= start && (int)lane < cnt && s_ == 1);
      const int e = smask ? (int)(__ffsll((long long)smask) - 1) : cnt;          // first split symbol of [start, cnt)
      if (e > start) {                                                          // run [start, e) without a split
        const bool act = (int)lane >= start && (int)lane < e;
        const int v_start = act ? UVOL_ALOAD(&vval[vn_]) : 0;
        int acc = 0;
        for (int k = start; k + 1 < e; k++) {
          const int ka = (int)UVOL_READLANE(va_, k), kn = (int)UVOL_READLANE(vn_, k), kp = (int)UVOL_READLANE(vp_, k);
          const uint32_t kd = UVOL_READLANE(dpk, k);
          const int hit = (vn_ == ka ? (int)(kd & 3u) : 0) + (vn_ == kn ? (int)((kd >> 2) & 3u) : 0) + (vn_ == kp ? (int)(kd >> 4) : 0);
          acc += (int)lane > k ? hit : 0;
        }
        if (act) {
          const int av = v_start - acc;
          if (mi > 0) { const int cv = av < 2 ? 2 : (av > 7 ? 7 : av); J.ctx_of[mi] = (uint8_t)(cv - 2); }
          if (dpk & 3u) UVOL_AADD(&vval[va_], -(int)(dpk & 3u));
          UVOL_AADD(&vval[vn_], -(int)((dpk >> 2) & 3u));
          UVOL_AADD(&vval[vp_], -(int)(dpk >> 4));
        }
        UVOL_WAVE_FENCE();
        UVOL_WAVE_SYNC();
      }
      if (e < cnt) {                                                            // the split symbol: serial, lane 0
        const int i = base + e;
        const int lc = (int)UVOL_READLANE(c_, e);
        const int ia = (int)UVOL_READLANE(va_, e), in_ = (int)UVOL_READLANE(vn_, e), ip = (int)UVOL_READLANE(vp_, e);
        if (lane == 0) {
          const int nx = g_nxt(lc), pv = g_prv(lc);
          const int val_n = UVOL_ALOAD(&vval[in_]), val_p = UVOL_ALOAD(&vval[ip]);
          UVOL_ASTORE(&vval[in_], val_n - 1); UVOL_ASTORE(&vval[ip], val_p - 1);
          int nleft = 0, a = opp[pv];
          while (a >= 0) { if (ftime[a / 3] <= i) break; nleft++; a = opp[g_nxt(a)]; }
          UVOL_ASTORE(&vval[ia], nleft + 1);
          const int newv = nvval; int nright = 0; a = opp[nx];
          while (a >= 0) { if (ftime[a / 3] <= i) break; nright++; c2vm[g_nxt(a)] = newv; a = opp[g_prv(a)]; }
          UVOL_ASTORE(&vval[nvval], nright + 1);
          if (i > 0) { const int cv = val_n < 2 ? 2 : (val_n > 7 ? 7 : val_n); J.ctx_of[i] = (uint8_t)(cv - 2); }
        }
        nvval++;
        // refresh the not-yet-consumed vertex ids of this chunk (corners right of the split now map to the new vertex)
        UVOL_WAVE_FENCE();
        UVOL_WAVE_SYNC();
        if (mi < nsym && (int)lane > e) { va_ = c2vm[c_]; vn_ = c2vm[g_nxt(c_)]; vp_ = c2vm[g_prv(c_)]; }
      }
      start = e + 1;
    }
  }
}

// symbols -> the six valence-context streams, in symbol order (wave ballots give each symbol its slot).  The streams lie back to back in
// ONE array of nf entries (six arrays of the worst-case length each were 4.8 MB per 200 k-face frame in flight): a first pass over the
// contexts counts, the second scatters.
__global__ void __launch_bounds__(64) k_eb_ctx(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.x];
  UVOL_SERIAL_PRIO();
  const uint32_t lane = threadIdx.x;
  const int nsym = J.status == 0 ? J.nsym : 0;
  uint32_t cnt[6] = {0, 0, 0, 0, 0, 0};
  for (int base = 1; base < nsym; base += 64) {
    const int i = base + (int)lane;
    const int cx = i < nsym ? J.ctx_of[i] : 7;
    for (int c = 0; c < 6; c++) cnt[c] += (uint32_t)__popcll(__ballot(cx == c));
  }
  uint32_t base_c[6]; { uint32_t o = 0; for (int c = 0; c < 6; c++) { base_c[c] = o; o += cnt[c]; } }
  uint32_t *all = J.ctx_all;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int base = 1; base < nsym; base += 64) {
    const int i = base + (int)lane;
    const bool in = i < nsym;
    const int cx = in ? J.ctx_of[i] : 7;
    const int ps = in ? J.symb[i - 1] : 0;
    const uint32_t id = ps == 0 ? 0u : (ps == 1 ? 1u : (ps == 3 ? 2u : (ps == 5 ? 3u : 4u)));
    for (int c = 0; c < 6; c++) {
      const unsigned long long m = __ballot(in && cx == c);
      if (in && cx == c) all[base_c[c] + (uint32_t)__popcll(m & lt)] = id;
      base_c[c] += (uint32_t)__popcll(m);
    }
  }
  if (lane == 0 && J.status == 0) for (int c = 0; c < 6; c++) { J.ctx_n[c] = cnt[c]; J.rs[c].n = cnt[c]; J.ctx_sym[c] = all + (base_c[c] - cnt[c]); J.rs[c].syms = J.ctx_sym[c]; }
}

// ------------------------------------------------------------------------------------------------
// Seams, seam bits and attribute vertices on the STORED tables (round 5).  Until round 4 the tables were renumbered into decoder
// order first (k_renumber_a / k_renumber_seams: five 3F-entry arrays written and read back, 48 MB of HBM traffic per 200 k-face frame,
// the largest kernel of the front end).  Nothing below needs that: a seam is a property of an edge, an attribute vertex a property of
// a fan, and the only order-dependent output - the seam bits, one per edge taken from the face with the LOWER decoder index, faces in
// decoder order, corners in decoder rotation - is produced by walking tstart[] (k_sb_count / k_sb_write).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool fseam_bit(const GeoJob &J, int slot, int c) { return (J.fseam[c / 3] >> (3 * slot + c % 3)) & 1u; }
// seam-masked opposite corner / swings on the stored table (slot < 0: the base table)
__device__ __forceinline__ int st_opp(const GeoJob &J, int slot, int c) { if (c < 0) return GEO_INV; if (slot >= 0 && fseam_bit(J, slot, c)) return GEO_INV; return J.opp[c]; }
__device__ __forceinline__ int st_swl(const GeoJob &J, int slot, int c) { const int o = st_opp(J, slot, g_nxt(c)); return o < 0 ? GEO_INV : g_nxt(o); }
__device__ __forceinline__ int st_swr(const GeoJob &J, int slot, int c) { const int o = st_opp(J, slot, g_prv(c)); return o < 0 ? GEO_INV : g_prv(o); }
__device__ __forceinline__ bool vseam_bit(const GeoJob &J, int slot, uint32_t v) { return (J.vseam[slot][v >> 5] >> (v & 31)) & 1u; }
// vertex of corner c in the table that sequences attribute slot `slot` (the base vertex unless an interior seam touches it)
__device__ __forceinline__ int att_vertex(const GeoJob &J, int slot, int c) {
  const int v = geo_vt(J)[c];
  return (J.interior_seams[slot] && vseam_bit(J, slot, (uint32_t)v)) ? J.avert[slot][c] : v;
}
// One thread per stored face: the seam flags of both attribute slots across its three edges (MeshAttributeCornerTable::InitFromAttribute:
// a boundary counts as a seam; an interior edge is one when the value ids at either end differ between the two faces), the 'an
// interior seam touches this vertex' bits and the per-slot 'has interior seams' flag.  The ids across an edge come from the face of the
// opposite corner: everything else is the face's own three 12-byte triples.
__global__ void __launch_bounds__(UVOL_BLOCK) k_seams(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f >= J.nf) return;
  const uvol_s3 o3 = *reinterpret_cast<const uvol_s3 *>(J.opp + 3 * (size_t)f);
  const int opp_[3] = { o3.x, o3.y, o3.z };
  uint32_t bits = 0;
  for (int i = 0; i < J.nad; i++) {
    const int32_t *A = J.att_kind[i] == 0 ? J.cu : J.cn;
    const uvol_s3 a3 = *reinterpret_cast<const uvol_s3 *>(A + 3 * (size_t)f);
    const int a[3] = { a3.x, a3.y, a3.z };
    bool any = false;
    for (int k = 0; k < 3; k++) {
      uint32_t sm = 1;
      if (opp_[k] >= 0) {
        const int oo = opp_[k], fo = 3 * (oo / 3), jo = oo - fo;
        const uvol_s3 b3 = *reinterpret_cast<const uvol_s3 *>(A + fo);     // the neighbour's three ids in one gather
        const int b[3] = { b3.x, b3.y, b3.z };
        sm = (a[(k + 1) % 3] != b[(jo + 2) % 3] || a[(k + 2) % 3] != b[(jo + 1) % 3]) ? 1u : 0u;
        if (sm) {                                                         // both ends of the edge get split
          any = true;
          const uint32_t va = (uint32_t)geo_vt(J)[3 * f + (k + 1) % 3], vb = (uint32_t)geo_vt(J)[3 * f + (k + 2) % 3];
          atomicOr(&J.vseam[i][va >> 5], 1u << (va & 31)); atomicOr(&J.vseam[i][vb >> 5], 1u << (vb & 31));
        }
      }
      bits |= sm << (3 * i + k);
    }
    if (any) J.interior_seams[i] = 1;
  }
  J.fseam[f] = (uint8_t)bits;
}
// seam bits, pass 1: one thread per DECODER-order face: which of its edges contribute a bit (the neighbour across it has the higher
// decoder index), in the decoder's corner rotation, and the bits themselves - one byte per face - plus the block sums for the scan
__global__ void __launch_bounds__(UVOL_BLOCK) k_sb_count(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  uint32_t cnt = 0;
  if (f < J.nf) {
    const int x0 = J.tstart[f], fo = x0 >> 2, r0 = x0 & 3;
    const uvol_s3 o3 = *reinterpret_cast<const uvol_s3 *>(J.opp + 3 * (size_t)fo);
    const int opp_[3] = { o3.x, o3.y, o3.z };
    const uint32_t fs = J.fseam[fo];
    int nb[3];
    for (int k = 0; k < 3; k++) nb[k] = opp_[k] < 0 ? -1 : J.face_time[opp_[k] / 3];
    uint32_t b0 = 0, b1 = 0;
    for (int k = 0; k < 3; k++) {
      const int j = (r0 + k) % 3;
      if (opp_[j] < 0) continue;
      const int t = nb[j], df = t >= 0 ? J.nsym - 1 - t : J.nsym + (-t - 2);
      if ((uint32_t)df <= f) continue;
      b0 |= ((fs >> j) & 1u) << cnt; b1 |= ((fs >> (3 + j)) & 1u) << cnt; cnt++;
    }
    J.sbpack[f] = (uint8_t)(cnt | (b0 << 2) | (b1 << 5));
  }
  const uint32_t tot = block_sum(cnt);
  if (threadIdx.x == 0 && blockIdx.x < uvol_blocks_dev(J.nf)) J.bsum[blockIdx.x] = tot;
}
// pass 2 (after k_scan_sums over SCAN_ELIG): the bits at their places, zero counts for the rabs coder
__global__ void __launch_bounds__(UVOL_BLOCK) k_sb_write(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  const bool ok = J.status == 0;
  const uint32_t nf = ok ? J.nf : 0u, f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const uint32_t pk = f < nf ? J.sbpack[f] : 0u, cnt = pk & 3u;
  uint32_t tot;
  const uint32_t pos = block_excl_scan(cnt, &tot) + ((ok && blockIdx.x <= uvol_blocks_dev(J.nf)) ? J.bsum[blockIdx.x] : 0);
  __shared__ uint32_t zc[2];
  if (threadIdx.x < 2) zc[threadIdx.x] = 0;
  __syncthreads();
  if (cnt) for (int i = 0; i < J.nad; i++) {
    const uint32_t b = (pk >> (2 + 3 * i)) & 7u; uint32_t z = 0;
    for (uint32_t k = 0; k < cnt; k++) { const uint8_t sb = (uint8_t)((b >> k) & 1u); J.seam_bits[i][pos + k] = sb; z += sb ? 0u : 1u; }
    if (z) atomicAdd(&zc[i], z);
  }
  __syncthreads();
  if (threadIdx.x < 2 && zc[threadIdx.x]) atomicAdd(&J.rb[1 + threadIdx.x].zeros, zc[threadIdx.x]);
  if (blockIdx.x == 0 && threadIdx.x == 0 && ok) {
    const uint32_t n = J.bsum[uvol_blocks_dev(J.nf)];
    J.n_elig = n; for (int i = 0; i < J.nad; i++) J.rb[1 + i].n = n;
  }
}

// attribute vertices of the vertices an interior seam touches (grid z = attribute slot): pass a gives every segment (maximal
// run of fan corners no seam / boundary separates) an id nverts_base + k at its left-most corner, pass b hands it to the other
// corners of the segment.  Corners of untouched vertices keep their base vertex and are NOT written (att_vertex).
__global__ void __launch_bounds__(UVOL_BLOCK) k_aseg_a(GeoJob *jobs) {
  JOB_OR_RETURN;
  const int i = (int)blockIdx.z;
  if (i >= J.nad || !J.interior_seams[i]) return;
  const uint32_t c0 = blockIdx.x * (UVOL_BLOCK * GEO_ILP) + threadIdx.x, nc = J.nc;
  int32_t v[GEO_ILP]; uint32_t w[GEO_ILP];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK; v[k] = c < nc ? geo_vt(J)[c] : 0; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) w[k] = J.vseam[i][(uint32_t)v[k] >> 5];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) {
    const uint32_t c = c0 + k * UVOL_BLOCK;
    if (c >= nc || !((w[k] >> ((uint32_t)v[k] & 31)) & 1u)) continue;
    // left-most corner of its segment <=> the edge to its left is a seam or a boundary <=> the seam flag of corner next(c)
    if (fseam_bit(J, i, g_nxt((int)c))) J.avert[i][c] = (int32_t)(J.nverts_t[0] + atomicAdd(&J.nseg[i], 1u));
  }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_aseg_b(GeoJob *jobs) {
  JOB_OR_RETURN;
  const int i = (int)blockIdx.z;
  if (i >= J.nad || !J.interior_seams[i]) return;
  const uint32_t c0 = blockIdx.x * (UVOL_BLOCK * GEO_ILP) + threadIdx.x, nc = J.nc;
  if (c0 == 0) { const uint32_t tot = J.nverts_t[0] + J.nseg[i]; J.nverts_t[2 + i] = tot; if (tot > J.ecap) J.status = GEO_E_WS_OVERFLOW; }
  uint32_t v[GEO_ILP], w[GEO_ILP];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK; v[k] = c < nc ? (uint32_t)geo_vt(J)[c] : 0u; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) w[k] = J.vseam[i][v[k] >> 5];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) {
    const uint32_t c = c0 + k * UVOL_BLOCK;
    if (c >= nc || !((w[k] >> (v[k] & 31)) & 1u)) continue;
    int l = (int)c; uint32_t guard = 0;
    for (;;) { const int nl = st_swl(J, i, l); if (nl < 0) break; l = nl; if (++guard > nc) { J.status = -22; return; } }
    if (l != (int)c) J.avert[i][c] = J.avert[i][l];
  }
}
