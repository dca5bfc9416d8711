// geo_conn.hpp - K4b: inverse maps, split events, valence replay, context streams, seams, attribute vertices.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
__global__ void __launch_bounds__(UVOL_BLOCK) k_face_time(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (i < (uint32_t)J.nsym) J.face_time[J.proc[i] / 3] = (int32_t)i;
}
// v2d[t][vertex] = position of the vertex in the coding order of table t: the inverse of order[t][] (same reason)
__global__ void __launch_bounds__(UVOL_BLOCK) k_v2d(GeoJob *jobs, int r8) {
  JOB_OR_RETURN;
  const int t = blockIdx.z;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (i >= J.ne[t]) return;
  if (t > 0 && (t - 1 >= J.nad || !J.interior_seams[t - 1])) return;
  const int c = J.order[t][i];
  const size_t code = (size_t)code_of_corner(c);
  int vi;
  if (r8 == 2) { const uint32_t *q = reinterpret_cast<const uint32_t *>(J.rec[1 + t]) + 4 * (size_t)(c / 3); vi = (int)((uint32_t)((((uint64_t)q[1] << 32) | q[0]) >> (21 * (c % 3))) & 0x1fffffu); }
  else vi = r8 ? (int)((uint32_t)J.rec[1 + t][2 * code] & 0x1fffffu) : J.rec[1 + t][4 * code];
  J.v2d[t][vi >> 1] = (int32_t)i;
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
    for (int k = 0; k < n; k++) { J.ev_src[pos + k] = (int)i; J.ev_spl[pos + k] = a[k]; J.ev_edge[pos + k] = (uint8_t)b[k]; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) J.nev = (int)J.bsum2[uvol_blocks_dev(J.nf)];
}

// working copies the replay mutates: corner -> vertex map (S symbols re-map corners to new vertices) and the valence per vertex
__global__ void __launch_bounds__(UVOL_BLOCK) k_valence_init(GeoJob *jobs) {
  JOB_OR_RETURN;
  const uint32_t t = blockIdx.x * UVOL_BLOCK + threadIdx.x, stride = gridDim.x * UVOL_BLOCK;
  { const uint32_t n4 = J.nc / 4;                                                     // 16 bytes per lane (both arrays are 16-byte aligned)
    const uint4 *src = reinterpret_cast<const uint4 *>(J.vert); uint4 *dst = reinterpret_cast<uint4 *>(J.c2vm);
    for (uint32_t q = t; q < n4; q += stride) dst[q] = src[q];
    for (uint32_t c = 4 * n4 + t; c < J.nc; c += stride) J.c2vm[c] = J.vert[c]; }
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

// symbols -> the six valence-context streams, in symbol order (wave ballots give each symbol its slot)
__global__ void __launch_bounds__(64) k_eb_ctx(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.x];
  UVOL_SERIAL_PRIO();
  const uint32_t lane = threadIdx.x;
  const int nsym = J.status == 0 ? J.nsym : 0;
  uint32_t base_c[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int base = 1; base < nsym; base += 64) {
    const int i = base + (int)lane;
    const bool in = i < nsym;
    const int cx = in ? J.ctx_of[i] : 7;
    const int ps = in ? J.symb[i - 1] : 0;
    const uint32_t id = ps == 0 ? 0u : (ps == 1 ? 1u : (ps == 3 ? 2u : (ps == 5 ? 3u : 4u)));
    for (int c = 0; c < 6; c++) {
      const unsigned long long m = __ballot(in && cx == c);
      if (in && cx == c) J.ctx_sym[c][base_c[c] + (uint32_t)__popcll(m & lt)] = id;
      base_c[c] += (uint32_t)__popcll(m);
    }
  }
  if (lane == 0 && J.status == 0) for (int c = 0; c < 6; c++) { J.ctx_n[c] = base_c[c]; J.rs[c].n = base_c[c]; }
}

// renumber into decoder order (decoder corner 3f+k <-> rot^k(processed corner f))
__device__ __forceinline__ int renum_first_corner(const GeoJob &J, uint32_t f) { return (int)f < J.nsym ? J.proc[J.nsym - 1 - (int)f] : J.initc[(int)f - J.nsym]; }
__global__ void __launch_bounds__(UVOL_BLOCK) k_renumber_a(GeoJob *jobs) {
  JOB_OR_RETURN;
  uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (f >= J.nf) return;
  const int c = renum_first_corner(J, f);
  const int o[3] = { c, g_nxt(c), g_prv(c) };
  for (int k = 0; k < 3; k++) J.new_of_old[o[k]] = (int)(3 * f + k);
}
// One thread per NEW face: the renumbered tables (opposite corners, value ids, vertices under the decoder's corner numbering),
// the attribute seams (MeshAttributeCornerTable::InitFromAttribute) with the seam-bit eligibility flags and their block sums,
// and the 'a seam touches this vertex' bits.  The renumbering maps whole faces (rotated), so everything a corner needs from
// its own face is in the thread's registers (three 12-byte loads per array from the OLD face) and the ids across an edge
// come from the old face of the opposite corner - 20 loads per face where the per-corner k_renumber_b + k_seams pair issued 60.
__global__ void __launch_bounds__(UVOL_BLOCK) k_renumber_seams(GeoJob *jobs) {
  JOB_OR_RETURN_UNIFORM;
  const uint32_t f = blockIdx.x * UVOL_BLOCK + threadIdx.x, nf = J.nf;
  const bool in = f < nf;
  __shared__ uint32_t ecnt[3];                                         // eligible corners per 256-corner block (three per 256 faces)
  if (threadIdx.x < 3) ecnt[threadIdx.x] = 0;
  __syncthreads();
  if (in) {
    const int c0 = renum_first_corner(J, f);
    const int fo = 3 * (c0 / 3), r0 = c0 - fo;                          // old face, rotation
    int opp_[3], P[3], U[3], Nn[3], V[3];
    { const uvol_s3 a = *reinterpret_cast<const uvol_s3 *>(J.opp + fo), b = *reinterpret_cast<const uvol_s3 *>(J.cp + fo), c = *reinterpret_cast<const uvol_s3 *>(J.cu + fo),
                    d = *reinterpret_cast<const uvol_s3 *>(J.cn + fo), e = *reinterpret_cast<const uvol_s3 *>(J.vert + fo);
      const int ao[3] = { a.x, a.y, a.z }, bo[3] = { b.x, b.y, b.z }, co[3] = { c.x, c.y, c.z }, dn[3] = { d.x, d.y, d.z }, ev[3] = { e.x, e.y, e.z };
      for (int k = 0; k < 3; k++) { const int j = (r0 + k) % 3; opp_[k] = ao[j]; P[k] = bo[j]; U[k] = co[j]; Nn[k] = dn[j]; V[k] = ev[j]; } }
    int no[3];
    for (int k = 0; k < 3; k++) no[k] = opp_[k] < 0 ? GEO_INV : J.new_of_old[opp_[k]];
    // ids across each edge: the two other corners of the opposite corner's OLD face
    int bu[3][2], bn[3][2];
    for (int k = 0; k < 3; k++) {
      const int oo = opp_[k] < 0 ? 0 : opp_[k];
      bu[k][0] = J.cu[g_prv(oo)]; bu[k][1] = J.cu[g_nxt(oo)]; bn[k][0] = J.cn[g_prv(oo)]; bn[k][1] = J.cn[g_nxt(oo)];
    }
    { uvol_s3 w; w.x = no[0]; w.y = no[1]; w.z = no[2]; *reinterpret_cast<uvol_s3 *>(J.nopp + 3 * (size_t)f) = w;
      w.x = P[0]; w.y = P[1]; w.z = P[2]; *reinterpret_cast<uvol_s3 *>(J.npid + 3 * (size_t)f) = w;
      w.x = U[0]; w.y = U[1]; w.z = U[2]; *reinterpret_cast<uvol_s3 *>(J.nuid + 3 * (size_t)f) = w;
      w.x = Nn[0]; w.y = Nn[1]; w.z = Nn[2]; *reinterpret_cast<uvol_s3 *>(J.nnid + 3 * (size_t)f) = w;
      w.x = V[0]; w.y = V[1]; w.z = V[2]; *reinterpret_cast<uvol_s3 *>(J.bvert + 3 * (size_t)f) = w; }
    for (int k = 0; k < 3; k++) {
      const uint32_t c = 3 * f + k; const bool e = no[k] >= 0 && (uint32_t)no[k] / 3 > f;
      J.elig[c] = e ? 1 : 0;
      if (e) atomicAdd(&ecnt[(3 * threadIdx.x + k) >> 8], 1u);
    }
    for (int i = 0; i < J.nad; i++) {
      const bool uvk = J.att_kind[i] == 0;
      bool any = false;
      for (int k = 0; k < 3; k++) {
        uint8_t sm = 1;
        if (opp_[k] >= 0) {
          const int a0 = uvk ? U[(k + 1) % 3] : Nn[(k + 1) % 3], a1 = uvk ? U[(k + 2) % 3] : Nn[(k + 2) % 3];
          const int b0 = uvk ? bu[k][0] : bn[k][0], b1 = uvk ? bu[k][1] : bn[k][1];
          sm = (a0 != b0 || a1 != b1) ? 1 : 0;
          if (sm) {                                                      // both ends of the edge get split
            any = true;
            const uint32_t va = (uint32_t)V[(k + 1) % 3], vb = (uint32_t)V[(k + 2) % 3];
            atomicOr(&J.vseam[i][va >> 5], 1u << (va & 31)); atomicOr(&J.vseam[i][vb >> 5], 1u << (vb & 31));
          }
        }
        J.seam[i][3 * (size_t)f + k] = sm;
      }
      if (any) J.interior_seams[i] = 1;
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) { const uint32_t b = 3 * blockIdx.x + threadIdx.x; if (b < uvol_blocks_dev(J.nc)) J.bsum[b] = ecnt[threadIdx.x]; }
}
// seam bits of the eligible corners, in corner order.  SB_E corners per thread (8-byte loads of the flags and of both seam
// arrays): with one corner per thread the kernel was 5 M workgroups per batch, each a chain of two byte loads and a scan
#define SB_E 8
__global__ void __launch_bounds__(UVOL_BLOCK) k_seam_bits(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  const bool ok = J.status == 0;
  const uint32_t nc = ok ? J.nc : 0u, c0 = (blockIdx.x * UVOL_BLOCK + threadIdx.x) * SB_E;
  unsigned long long e8 = 0;
  if (c0 + SB_E <= nc) e8 = *reinterpret_cast<const unsigned long long *>(J.elig + c0);
  else for (uint32_t k = 0; c0 + k < nc && k < SB_E; k++) e8 |= (unsigned long long)J.elig[c0 + k] << (8 * k);
  e8 &= 0x0101010101010101ull;
  uint32_t cnt = (uint32_t)__popcll(e8), tot;
  uint32_t pos = block_excl_scan(cnt, &tot) + ((ok && blockIdx.x * SB_E <= uvol_blocks_dev(J.nc)) ? J.bsum[blockIdx.x * SB_E] : 0);
  // zero counts: one global atomic per block and attribute
  __shared__ uint32_t zc[2];
  if (threadIdx.x < 2) zc[threadIdx.x] = 0;
  __syncthreads();
  if (cnt) for (int i = 0; i < J.nad; i++) {
    unsigned long long s8 = 0;
    if (c0 + SB_E <= nc) s8 = *reinterpret_cast<const unsigned long long *>(J.seam[i] + c0);
    else for (uint32_t k = 0; c0 + k < nc && k < SB_E; k++) s8 |= (unsigned long long)J.seam[i][c0 + k] << (8 * k);
    uint32_t p = pos, z = 0;
    for (int k = 0; k < SB_E; k++) if ((e8 >> (8 * k)) & 1ull) { const uint8_t sb = (uint8_t)(s8 >> (8 * k)); J.seam_bits[i][p++] = sb; z += sb ? 0u : 1u; }
    if (z) atomicAdd(&zc[i], z);
  }
  __syncthreads();
  if (threadIdx.x < 2 && zc[threadIdx.x]) atomicAdd(&J.rb[1 + threadIdx.x].zeros, zc[threadIdx.x]);
  if (blockIdx.x == 0 && threadIdx.x == 0 && ok) {
    uint32_t n = J.bsum[uvol_blocks_dev(J.nc)];
    J.n_elig = n; for (int i = 0; i < J.nad; i++) J.rb[1 + i].n = n;
  }
}

// attribute vertices of the vertices an interior seam touches (grid z = attribute slot): pass a gives every segment (maximal
// run of fan corners no seam / boundary separates) an id nverts_base + k at its left-most corner, pass b hands it to the other
// corners of the segment; all other corners keep their base vertex
__global__ void __launch_bounds__(UVOL_BLOCK) k_aseg_a(GeoJob *jobs) {
  JOB_OR_RETURN;
  const int i = (int)blockIdx.z;
  if (i >= J.nad || !J.interior_seams[i]) return;
  const uint32_t c0 = blockIdx.x * (UVOL_BLOCK * GEO_ILP) + threadIdx.x, nc = J.nc;
  int32_t v[GEO_ILP]; uint32_t w[GEO_ILP];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK; v[k] = c < nc ? J.bvert[c] : 0; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) w[k] = J.vseam[i][(uint32_t)v[k] >> 5];
  // left-most corner of its segment <=> the edge to its left is a seam or a boundary <=> seam[next(c)] (k_seams marks boundaries
  // too); fetched for every corner (a neighbouring byte) so that the rare seam vertices cost no divergent round trip
  uint8_t sl[GEO_ILP];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK; sl[k] = J.seam[i][g_nxt(c < nc ? c : 0u)]; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) {
    const uint32_t c = c0 + k * UVOL_BLOCK;
    if (c >= nc) continue;
    if (!((w[k] >> ((uint32_t)v[k] & 31)) & 1u)) { J.avert[i][c] = v[k]; continue; }
    if (sl[k]) J.avert[i][c] = (int32_t)(J.nverts_t[0] + atomicAdd(&J.nseg[i], 1u));
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
  for (int k = 0; k < GEO_ILP; k++) { const uint32_t c = c0 + k * UVOL_BLOCK; v[k] = c < nc ? (uint32_t)J.bvert[c] : 0u; }
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) w[k] = J.vseam[i][v[k] >> 5];
  GTab T; T.opp = J.nopp; T.seam = J.seam[i];
#pragma unroll
  for (int k = 0; k < GEO_ILP; k++) {
    const uint32_t c = c0 + k * UVOL_BLOCK;
    if (c >= nc || !((w[k] >> (v[k] & 31)) & 1u)) continue;
    int l = (int)c; uint32_t guard = 0;
    for (;;) { const int nl = gt_swl(T, l); if (nl < 0) break; l = nl; if (++guard > nc) { J.status = -22; return; } }
    if (l != (int)c) J.avert[i][c] = J.avert[i][l];
  }
}

