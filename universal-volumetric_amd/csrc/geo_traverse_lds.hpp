// geo_traverse_lds.hpp - K5: DepthFirstTraverser, LDS (wave-per-walker) forms.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
// ------------------------------------------------------------------------------------------------
// K5: DepthFirstTraverser — serial per (table, frame), one lane each.  t=0 base table, t=1,2 attribute tables.
// One record load per face (RecOps); visited faces / vertices are bitmaps in LDS; order[] is the only output stream
// (v2d[], its inverse, is rebuilt by k_v2d).  Same structure as eb_walk_lane0.
// ------------------------------------------------------------------------------------------------
template <bool R8, typename FB, typename VB>
__device__ __forceinline__ void traverse_lane0(GeoJob &J, int t, FB fbits, VB vbits, UVOL_L(uint32_t) stg_lds) {
  typedef RecOps<R8> RO;
  const int nf = (int)J.nf;
  const typename RO::Ptr rec = RO::ptr(J.rec[1 + t]);
  UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.t_stack[t]); UVOL_G(int32_t) order = UVOL_TO_G(int32_t, J.order[t]);
  const int dz = UVOL_LANE_ZERO();
  WalkStage stg; stg.init(stg_lds);
  int n = 0;
#define T_EMIT(C) do { stg.w[n & (WALK_STG - 1)] = (C); n++; if ((n & (WALK_STG - 1)) == 0) stg.flush_words(order, n); } while (0)
  UVOL_G(const int32_t) tstart = UVOL_TO_G(const int32_t, J.tstart);
  const bool virt = J.tstart != nullptr;                  // encode side: components start in DECODER order, tstart[f] = first corner of decoder face f
  int nvis = 0;
  for (int f = 0; f < nf; f++) {
    int x;
    if (virt) { if (nvis >= nf) break; x = tstart[f]; }
    else { if ((f & 31) == 0) { while (f + 32 <= nf && pword(fbits, f >> 5) == 0xffffffffu) f += 32; if (f >= nf) break; } x = 4 * f; }
    if (pbit_get(fbits, x >> 2)) continue;
    int sp = 0;
    stack[sp] = x;
    sp++;
    int top = x; bool top_known = true;
    { const int xn = code_nxt(x), xp = code_prv(x);
      int vn, vp, r_, l_; RO::get(rec, xn, vn, r_, l_); RO::get(rec, xp, vp, r_, l_); vn >>= 1; vp >>= 1;
      if (!pbit_get(vbits, vn)) { pbit_set(vbits, vn); T_EMIT(corner_of_code(xn)); }
      if (!pbit_get(vbits, vp)) { pbit_set(vbits, vp); T_EMIT(corner_of_code(xp)); } }
    while (sp > 0) {
      x = top_known ? top : stack[sp - 1];
      top_known = false;
      if (x < 0 || pbit_get(fbits, x >> 2)) { sp--; continue; }
      int vi, rc, lc;
      RO::get(rec, x, vi, rc, lc);
      for (;;) {
        const int face = x >> 2;
        // both records this step can move to are requested now and taken (readfirstlane) only by the branch that goes there
        const typename RO::Pre pR = RO::pre(rec, (rc < 0 ? x : rc) + dz), pL = RO::pre(rec, (lc < 0 ? x : lc) + dz);
        pbit_set(fbits, face);
        nvis++;
        const int v = vi >> 1;
        // the three bitmap words this step can need, read together (one LDS round trip)
        const uint32_t vw_ = pword(vbits, v >> 5);
        const uint32_t rw_ = rc < 0 ? 0xffffffffu : pword(fbits, rc >> 7), lw_ = lc < 0 ? 0xffffffffu : pword(fbits, lc >> 7);
#define T_GO_R() do { x = rc; RO::take(pR, vi, rc, lc); } while (0)
#define T_GO_L() do { x = lc; RO::take(pL, vi, rc, lc); } while (0)
        if (!((vw_ >> (v & 31)) & 1u)) {
          pbit_set(vbits, v); T_EMIT(3 * face + (x & 3));
          if (!(vi & 1)) { T_GO_R(); continue; }
        }
        const bool rvis = ((rw_ >> ((rc >> 2) & 31)) & 1u) != 0, lvis = ((lw_ >> ((lc >> 2) & 31)) & 1u) != 0;
        if (rvis) { if (lvis) { sp--; break; } T_GO_L(); }
        else { if (lvis) T_GO_R(); else { stack[sp - 1] = lc; stack[sp] = rc; sp++; top = rc; top_known = true; break; } }
#undef T_GO_R
#undef T_GO_L
      }
    }
  }
#undef T_EMIT
  stg.tail_words(order, n);
  J.ne[t] = (uint32_t)n;
  if (t == 0 && J.nverts != 0xffffffffu && (uint32_t)n != J.nverts) J.status = -11;      // (the decode path has no expected count)
}


// cooperative-lane form of traverse_lane0 (see eb_walk_coop): same traversal, same order[] stream
template <bool R8>
__device__ __forceinline__ void traverse_coop(GeoJob &J, int t, UVOL_L(uint32_t) lds, uint32_t fw, UVOL_L(uint32_t) pub, int pf) {
  typedef RecOps<R8> RO;
  const int lane = (int)(threadIdx.x & 63);
  const bool cl = lane < 2;
  const int nf = (int)J.nf;
  const typename RO::Ptr rec = RO::ptr(J.rec[1 + t]);
  UVOL_G(int32_t) stack = UVOL_TO_G(int32_t, J.t_stack[t]); UVOL_G(int32_t) order = UVOL_TO_G(int32_t, J.order[t]);
  UVOL_L(uint32_t) dummy = pub + 4 + (lane & 1);
  uint32_t ov = 0;                                       // staged order[] entries: lane k = entry (n & ~63) + k
  int n = 0;
#define C_FWORD(k) ((uint32_t)UVOL_BCAST0(lds[k]))
#define C_EMIT(C) do { ov = UVOL_WRITELANE((C), n & 63, ov); n++; if ((n & 63) == 0) order[n - 64 + lane] = (int32_t)ov; } while (0)
  const int32_t *tstart = J.tstart;
  const bool virt = tstart != nullptr;                    // encode side: components start in DECODER order, tstart[f] = first corner of decoder face f
  int nvis = 0;
  for (int f = 0; f < nf; f++) {
    int x;
    if (virt) {
      // the next decoder-order face that is not visited yet: 64 candidates per round trip (every lane tests one), the first by a ballot
      if (nvis >= nf) break;
      bool found = false;
      while (f < nf) {
        const int fk = f + lane, xk = fk < nf ? tstart[fk] : -1;
        const bool un = xk >= 0 && !((lds[xk >> 7] >> ((xk >> 2) & 31)) & 1u);
        const unsigned long long m = __ballot(un);
        if (m) { const int k0 = (int)__ffsll((long long)m) - 1; f += k0; x = (int)UVOL_READLANE(xk, k0); found = true; break; }
        f += 64;
      }
      if (!found) break;
    } else {
      if ((f & 31) == 0) { while (f + 32 <= nf && C_FWORD(f >> 5) == 0xffffffffu) f += 32; if (f >= nf) break; }
      if ((C_FWORD(f >> 5) >> (f & 31)) & 1u) continue;
      x = 4 * f;
    }
    int sp = 0;
    if (lane == 0) stack[sp] = x;
    sp++;
    int top = x; bool top_known = true;
    { const int xn = code_nxt(x), xp = code_prv(x);
      int vn, vp, r_, l_; coop_get<R8>(rec, xn, vn, r_, l_); coop_get<R8>(rec, xp, vp, r_, l_); vn >>= 1; vp >>= 1;
      uint32_t w = (uint32_t)UVOL_BCAST0(lds[fw + (vn >> 5)]);
      if (!((w >> (vn & 31)) & 1u)) { lds[fw + (vn >> 5)] = w | (1u << (vn & 31)); C_EMIT(corner_of_code(xn)); }
      w = (uint32_t)UVOL_BCAST0(lds[fw + (vp >> 5)]);
      if (!((w >> (vp & 31)) & 1u)) { lds[fw + (vp >> 5)] = w | (1u << (vp & 31)); C_EMIT(corner_of_code(xp)); } }
    while (sp > 0) {
      if (top_known) x = top; else { UVOL_WAVE_FENCE(); x = UVOL_BCAST0(stack[sp - 1]); }
      top_known = false;
      if (x < 0) { sp--; continue; }
      uint32_t xw = C_FWORD(x >> 7);
      if ((xw >> ((x >> 2) & 31)) & 1u) { sp--; continue; }
      int vi, rc, lc;
      coop_get<R8>(rec, x, vi, rc, lc);
      for (;;) {                                          // straight-line step, see eb_walk_coop
        const int face = x >> 2;
        const int cand = lane == 0 ? rc : lc; const bool cvalid = cand >= 0;
        const int ccode = cvalid ? cand : x;
        const typename RO::Pre pre = RO::pre(rec, ccode);
        if (pf) pub[0] = (uint32_t)x;
        lds[face >> 5] = xw | (1u << (face & 31));
        nvis++;
        const int v = vi >> 1;
        const uint32_t widx = cl ? (uint32_t)ccode >> 7 : fw + (uint32_t)(v >> 5);
        const uint32_t sh = cl ? ((uint32_t)cand >> 2) & 31u : (uint32_t)v & 31u;
        const uint32_t word = lds[widx];
        const bool hit = ((word >> sh) & 1u) != 0 || (cl && !cvalid);
        const uint32_t m = (uint32_t)__ballot(hit) & 7u;
        (cl ? dummy : lds + widx)[0] = word | (1u << sh);
        // a vertex seen for the first time takes the next place in the order (the slot is simply overwritten otherwise)
        ov = UVOL_WRITELANE(3 * face + (x & 3), n & 63, ov);
        const int fresh = (int)((m >> 2) & 1u) ^ 1;
        n += fresh;
        if (__builtin_expect(fresh && (n & 63) == 0, 0)) order[n - 64 + lane] = (int32_t)ov;
        const bool ccase = (((m >> 2) | (uint32_t)vi) & 1u) == 0;
        const uint32_t k = ccase ? 0u : 1u + (m & 3u);   // 0: go right (new interior vertex); 1: fork; 2: right visited -> left; 3: left visited -> right; 4: dead end
        if (__builtin_expect((0x12u >> k) & 1u, 0)) {                              // fork (1) or dead end (4)
          if (k == 4u) { sp--; break; }
          if (lane == 0) { stack[sp - 1] = lc; stack[sp] = rc; }
          sp++; top = rc; top_known = true; break;
        }
        const int sel = k == 2u ? 1 : 0;
        x = sel ? lc : rc;
        xw = UVOL_READLANE(word, sel);
        CoopRec<R8>::take(pre, sel, vi, rc, lc);
      }
    }
  }
#undef C_EMIT
#undef C_FWORD
  if (lane < (n & 63)) order[(n & ~63) + lane] = (int32_t)ov;
  if (lane == 0) {
    J.ne[t] = (uint32_t)n;
    if (t == 0 && J.nverts != 0xffffffffu && (uint32_t)n != J.nverts) J.status = -11;
  }
}

template <bool R8>
__global__ void __launch_bounds__(128) k_traverse(GeoJob *jobs, int vcap_words, int dbg) {
  GeoJob &J = jobs[blockIdx.y];
  const int t = blockIdx.x;
  UVOL_SERIAL_PRIO();
  UVOL_DYN_SMEM(uint32_t, lds);
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  const int ai = t > 0 ? t - 1 : 0;
  const bool ok = J.status == 0 && !(t > 0 && (ai >= J.nad || !J.interior_seams[ai]));
  const uint32_t fw = ((uint32_t)J.nf + 31) / 32, vw = (J.nverts_t[1 + t] + 31) / 32, vcw = (uint32_t)vcap_words;
  const bool v_in_lds = vw <= vcw;
  const uint32_t stg_off = (fw + vcw + 3u) & ~3u;
  if (ok) for (uint32_t k = tid; k < fw + vcw; k += 128) lds[k] = 0;
  if (tid < WALK_PUB_DWORDS) lds[stg_off + WALK_STG_DWORDS + tid] = 0;
  __syncthreads();
  if (!ok) return;
  UVOL_L(uint32_t) stg = UVOL_TO_L(uint32_t, lds) + stg_off; UVOL_L(uint32_t) pub = stg + WALK_STG_DWORDS;
  if (v_in_lds) {
    if (wave == 1) { if (dbg & 2) walk_prefetch_wave(J.rec[1 + t], (uint32_t)((R8 ? 32 : 64) * (size_t)J.nf), pub, R8 ? 4 : 3); return; }
    traverse_coop<R8>(J, t, UVOL_TO_L(uint32_t, lds), fw, pub, dbg & 2);
    pub[1] = 1u;
    return;
  }
  if (tid != 0) return;
  traverse_lane0<R8>(J, t, UVOL_TO_L(uint32_t, lds), UVOL_TO_G(uint32_t, reinterpret_cast<uint32_t *>(J.t_vvis[t])), stg);
}

