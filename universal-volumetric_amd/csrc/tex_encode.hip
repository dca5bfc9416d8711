// tex_encode.hip — hand-written HIP (gfx950) texture encoder: B RGBA8 layers -> one ETC1S/BasisLZ .ktx2.
//
// Replaces the arithmetic of HOT LOOP 2 of the reference (scripts/Encoder.py:279-298, one
// `basisu -ktx2 -tex_type video -multifile_num B -y_flip` process per KTX2_BATCH_SIZE images).
// Kernel groups: K8 tile load + y-flip, K9 ETC1S endpoint search, K10 endpoint /
// selector codebook clustering, K11 P-frame skip decision, K12 BasisLZ slice packing (Huffman build,
// prefix-scan of code lengths, atomicOr bit packing); K13 (container) is host code at the bottom.
// The algorithm is the deterministic integer one documented in DESIGN.md §texture; it produces
// byte-identical files to the CPU restatement used by the tests.
//
// No MFMA: per-block integer SSE search, integer VQ statistics (64-bit atomics), Huffman bit work.
#include "uvol_common.hpp"
#include "tex_device.hpp"
#include <algorithm>

#define TJOB_OR_RETURN TexJob &J = job[blockIdx.z]; if (J.status != 0) return
#define TEX_RETRY_ALPHA (~(size_t)0)          // out_lens[] marker between the two passes of tex_encode_segments

// ------------------------------------------------------------------------------------------------
// scans (block-level exclusive scan shared with nothing else in this TU)
// ------------------------------------------------------------------------------------------------
__device__ inline uint32_t t_block_excl_scan(uint32_t v, uint32_t *total) {
  __shared__ uint32_t wsum[UVOL_BLOCK / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t x = v;
  for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int i = 0; i < UVOL_BLOCK / 64; i++) { if (i < w) base += wsum[i]; tot += wsum[i]; }
  __syncthreads();
  *total = tot;
  return base + x - v;
}
// phase 1: per-block sums of flag[0..n)
__global__ void __launch_bounds__(UVOL_BLOCK) k_tscan_a(TexJob *job, uint32_t n) {
  TexJob &J = job[blockIdx.z];
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  uint32_t v = (J.status == 0 && i < n) ? J.flag[i] : 0, tot;
  t_block_excl_scan(v, &tot);
  if (threadIdx.x == 0) J.bsum[blockIdx.x] = tot;
}
// phase 2: exclusive scan of the block sums (single workgroup); bsum[nblocks] = grand total
__global__ void __launch_bounds__(UVOL_BLOCK) k_tscan_b(TexJob *job, uint32_t nblocks) {
  TexJob &J = job[blockIdx.z];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nblocks; b0 += UVOL_BLOCK) {
    const uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < nblocks ? J.bsum[i] : 0, tot;
    const uint32_t ex = t_block_excl_scan(v, &tot);
    const uint32_t c = carry;
    if (i < nblocks) J.bsum[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) J.bsum[nblocks] = carry;
}

// ------------------------------------------------------------------------------------------------
// K11: P-frame skip flags — one thread per block position walks the layers against its anchor block
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(UVOL_BLOCK) k_tex_skip(TexJob *job) {
  TJOB_OR_RETURN;
  const uint32_t b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (b >= J.nb) return;
  const uint32_t X = b % J.bx, Y = b / J.bx;
  uint32_t anchor[16], cur[16], amask = 0xff000000u;
  const uint32_t st = 1u << J.ashift;                     // slices of one kind follow each other at this distance
  for (uint32_t kind = 0; kind < st; kind++) {
    t_load_block(J, kind, X, Y, anchor, &amask);
    J.skip[(size_t)kind * J.nb + b] = 0; J.flag[(size_t)kind * J.nb + b] = 1;
    for (uint32_t l = kind + st; l < J.L; l += st) {
      t_load_block(J, l, X, Y, cur, &amask);
      uint32_t d = 0;
      for (int i = 0; i < 16; i++) {
        const int dr = (int)(cur[i] & 255) - (int)(anchor[i] & 255), dg = (int)((cur[i] >> 8) & 255) - (int)((anchor[i] >> 8) & 255), db = (int)((cur[i] >> 16) & 255) - (int)((anchor[i] >> 16) & 255);
        d += (uint32_t)(dr * dr + dg * dg + db * db);
      }
      const bool sk = d <= J.T_skip;
      J.skip[(size_t)l * J.nb + b] = sk ? 1 : 0; J.flag[(size_t)l * J.nb + b] = sk ? 0 : 1;      // coded blocks -> item list (k_item_compact)
      if (!sk) for (int i = 0; i < 16; i++) anchor[i] = cur[i];
    }
  }
  // an image with alpha != 255 met in the opaque layout: the host encodes this segment again with alpha slices (every block of
  // every layer passes through here)
  if (!J.ashift && (amask & 0xff000000u) != 0xff000000u) J.status = TEX_E_ALPHA;
}

// ------------------------------------------------------------------------------------------------
// K8+K9: tile load + per-block ETC1S endpoint fit (colour5 + intensity table), histogram of the fits
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(UVOL_BLOCK) k_tex_fit(TexJob *job) {
  TJOB_OR_RETURN;
  // one lane per CODED block (item list built right after the skip decision): with ~56 % of a video segment's blocks skipped,
  // a grid over all blocks ran this — the most expensive texture kernel — with fewer than half of its lanes active
  const uint32_t it = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (it >= J.n_items) return;
  const uint32_t b = J.item[it];
  const uint32_t l = b / J.nb, r = b % J.nb;
  uint32_t px[16]; t_load_block(J, l, r % J.bx, r / J.bx, px);
  int sum[3] = {0, 0, 0};
  for (int i = 0; i < 16; i++) { sum[0] += (int)(px[i] & 255); sum[1] += (int)((px[i] >> 8) & 255); sum[2] += (int)((px[i] >> 16) & 255); }
  int base5[3]; for (int c = 0; c < 3; c++) base5[c] = (((sum[c] + 8) >> 4) * 31 + 127) / 255;
  uint32_t best = 0xffffffffu; int bc0 = 0, bc1 = 0, bc2 = 0, bt = 0;
  for (int t = 0; t < 8; t++) for (int di = 0; di < 3; di++) {
    const int dd = di == 0 ? 0 : (di == 1 ? -1 : 1);
    const int c0 = t_clampi(base5[0] + dd, 0, 31), c1 = t_clampi(base5[1] + dd, 0, 31), c2 = t_clampi(base5[2] + dd, 0, 31);
    const uint32_t e = t_eval_block<false, false>(px, c0, c1, c2, t, nullptr, nullptr);
    if (e < best) { best = e; bt = t; bc0 = c0; bc1 = c1; bc2 = c2; }
  }
  for (int c = 0; c < 3; c++) for (int di = 1; di < 3; di++) {
    const int dd = di == 1 ? -1 : 1;
    int c0 = bc0 + (c == 0 ? dd : 0), c1 = bc1 + (c == 1 ? dd : 0), c2 = bc2 + (c == 2 ? dd : 0);
    const int cc = c == 0 ? c0 : (c == 1 ? c1 : c2);
    if (cc < 0 || cc > 31) continue;
    const uint32_t e = t_eval_block<false, false>(px, c0, c1, c2, bt, nullptr, nullptr);
    if (e < best) { best = e; bc0 = c0; bc1 = c1; bc2 = c2; }
  }
  const uint32_t cell = ((uint32_t)bt << 15) | ((uint32_t)bc0 << 10) | ((uint32_t)bc1 << 5) | (uint32_t)bc2;
  J.cell[b] = cell;
  atomicAdd(&J.hist[cell], 1u);
}

// non-empty histogram cells -> compact, ascending list (cid, cw) + inverse map
__global__ void __launch_bounds__(UVOL_BLOCK) k_cell_flags(TexJob *job) {
  TJOB_OR_RETURN;
  const uint32_t c = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (c < (1u << 18)) J.flag[c] = J.hist[c] ? 1 : 0;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_cell_compact(TexJob *job) {
  TexJob &J = job[blockIdx.z];
  const uint32_t c = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const bool live = J.status == 0 && c < (1u << 18);
  uint32_t v = live ? J.flag[c] : 0, tot;
  const uint32_t pos = t_block_excl_scan(v, &tot) + J.bsum[blockIdx.x];
  if (live && v) { J.cid[pos] = c; J.cw[pos] = J.hist[c]; J.cidx[c] = pos; }
  if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) {
    const uint32_t n = J.bsum[(1u << 18) / UVOL_BLOCK];
    J.ncell = n;
    TexVQ &V = J.vq[0]; V.n_items = n; V.K = n < J.Kmax_e ? n : J.Kmax_e; V.nl = 1; V.done = (V.K <= 1) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// K10: level-synchronous tree-structured VQ.  DIM=4: items = histogram cells (weighted);
//      DIM=16: items = coded blocks' selector vectors.
// ------------------------------------------------------------------------------------------------
template <int DIM> __device__ __forceinline__ void vq_item(const TexJob &J, uint32_t i, int x[DIM], unsigned long long &w) {
  if (DIM == 4) { t_cell_coords(J.cid[i], x); w = J.cw[i]; }
  else { const uint32_t s = J.bsel[J.item[i]]; for (int k = 0; k < DIM; k++) x[k] = (int)((s >> (2 * k)) & 3); w = 1; }
}
template <int DIM> __device__ __forceinline__ int vq_wd(int d) { return (DIM == 4 && d == 3) ? 2 : 1; }

template <int DIM>
__global__ void __launch_bounds__(UVOL_BLOCK) k_vq_zero(TexJob *job, int force) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[DIM == 4 ? 0 : 1];
  if (V.done && !force) return;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (i < V.K) V.stW[i] = 0;
  if (i < V.K * DIM) { V.stS[i] = 0; V.stQ[i] = 0; }
}
// leaf statistics: LDS-privatised counters (CT = u64 for weighted cells, u32 for unit-weight selector
// vectors whose per-workgroup partial sums cannot overflow) for the first LCAP leaves, global atomics beyond
// (leaf_base: the statistics of leaves [leaf_base, leaf_base + LCAP) — one pass per LCAP leaves of the codebook, so no
//  leaf ever falls back to contended global atomics; a pass whose range is beyond the current leaf count returns at once)
template <int DIM, int LCAP, typename CT>
__global__ void __launch_bounds__(UVOL_BLOCK) k_vq_stats(TexJob *job, int force, uint32_t leaf_base, uint32_t lds_leaves) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[DIM == 4 ? 0 : 1];
  if ((V.done && !force) || V.nl <= leaf_base) return;
  UVOL_DYN_SMEM(CT, lds);                       // [lds_leaves * (1 + 2*DIM)], lds_leaves <= LCAP (round r has <= 2^r leaves)
  const uint32_t nl = V.nl - leaf_base, pass = nl < (uint32_t)LCAP ? nl : (uint32_t)LCAP, ncap = pass < lds_leaves ? pass : lds_leaves, stride = 1 + 2 * DIM;
  for (uint32_t k = threadIdx.x; k < ncap * stride; k += UVOL_BLOCK) lds[k] = 0;
  __syncthreads();
  for (uint32_t base = blockIdx.x * UVOL_BLOCK; base < V.n_items; base += gridDim.x * UVOL_BLOCK) {
    const uint32_t i = base + threadIdx.x;
    const bool todo = i < V.n_items;
    const uint32_t l = todo ? V.leaf[i] : 0xffffffffu;
    if (!todo || l < leaf_base || l - leaf_base >= pass) continue;          // other leaves: another pass
    int x[DIM]; unsigned long long w; vq_item<DIM>(J, i, x, w);
    if (l - leaf_base >= ncap) {                                            // not expected (host sizes the LDS for the round): stay correct
      atomicAdd(&V.stW[l], w);
      for (int d = 0; d < DIM; d++) { atomicAdd(&V.stS[(size_t)l * DIM + d], w * (unsigned long long)x[d]); atomicAdd(&V.stQ[(size_t)l * DIM + d], w * (unsigned long long)(x[d] * x[d])); }
      continue;
    }
    CT *p = lds + (size_t)(l - leaf_base) * stride;
    atomicAdd(&p[0], (CT)w);
    for (int d = 0; d < DIM; d++) { atomicAdd(&p[1 + d], (CT)(w * (unsigned long long)x[d])); atomicAdd(&p[1 + DIM + d], (CT)(w * (unsigned long long)(x[d] * x[d]))); }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < ncap * stride; k += UVOL_BLOCK) {
    const unsigned long long v = (unsigned long long)lds[k]; if (!v) continue;
    const uint32_t l = leaf_base + k / stride, f = k % stride;
    if (f == 0) atomicAdd(&V.stW[l], v);
    else if (f <= (uint32_t)DIM) atomicAdd(&V.stS[(size_t)l * DIM + (f - 1)], v);
    else atomicAdd(&V.stQ[(size_t)l * DIM + (f - 1 - DIM)], v);
  }
}
// Selector-vector statistics (DIM 16, unit weights): 17 LDS words per leaf {S0|Q0, S1|Q1, ... , S15|Q15, W}, S in the low and Q
// in the high half of a word.  One pass privatises the lcap <= 256 leaves from leaf_base (34 KiB of LDS with 64-bit words);
// the host runs one pass per 256 leaves the round can have.
#define SEL_ILP 4
// one wave's items into the LDS table: `todo` items carry the window-relative leaf `rl` (< ncap) and the selector word `sw`.
// Table layout: 17 words per leaf, word d < 16 = S_d | Q_d << 16, word 16 = W.
template <typename WT> __device__ __forceinline__ void sel_stats_accumulate(WT *lds, bool todo, uint32_t rl, uint32_t sw, uint32_t lane) {
  constexpr int QS = sizeof(WT) * 4;            // S in the low half of a word, Q in the high half
  // While a large part of the wave sits in one leaf (always in the early rounds) count with ballots - c_v = popc(ballot(x_d == v)),
  // S = c1+2c2+3c3, Q = c1+4c2+9c3 - and let 17 lanes post one add each.  Smaller groups take the per-item path below.
  unsigned long long rem = __ballot(todo);
  for (int rounds = 0; rounds < 4 && rem; rounds++) {
    const uint32_t leader = (uint32_t)(__ffsll((long long)rem) - 1);
    const uint32_t ll = UVOL_READLANE(rl, leader);
    const bool inm = todo && rl == ll;
    const unsigned long long m = __ballot(inm);
    if (__popcll(m) < 24) break;
    WT myv = 0;
    for (int d = 0; d < 16; d++) {
      const uint32_t xv = (sw >> (2 * d)) & 3u;
      const uint32_t c1 = (uint32_t)__popcll(__ballot(inm && xv == 1)), c2 = (uint32_t)__popcll(__ballot(inm && xv == 2)), c3 = (uint32_t)__popcll(__ballot(inm && xv == 3));
      if (lane == (uint32_t)d) myv = (WT)(c1 + 2 * c2 + 3 * c3) | ((WT)(c1 + 4 * c2 + 9 * c3) << QS);
    }
    if (lane == 16) myv = (WT)__popcll(m);
    if (lane < 17 && myv) atomicAdd(&lds[ll * 17 + lane], myv);
    rem &= ~m;
    if (inm) todo = false;
  }
  if (!todo) return;
  // per item: 17 adds, each lane starting at its own word (lane mod 17) - lanes of one leaf then hit different words in the
  // same instruction instead of queueing on one address (same-address LDS atomics of a wave are serialised)
  WT *p = lds + (size_t)rl * 17;
  uint32_t f = lane % 17u;
  for (int k = 0; k < 17; k++) {
    const uint32_t xv = (sw >> (2 * (f & 15u))) & 3u;
    const WT v = f == 16u ? (WT)1 : ((WT)xv | ((WT)(xv * xv) << QS));
    if (v) atomicAdd(&p[f], v);
    f = f == 16u ? 0u : f + 1u;
  }
}
template <typename WT> __device__ __forceinline__ void sel_stats_flush(TexVQ &V, const WT *lds, uint32_t ncap, uint32_t leaf0) {
  constexpr int QS = sizeof(WT) * 4;
  const WT lo = ((WT)1 << QS) - 1;
  for (uint32_t k = threadIdx.x; k < ncap * 17; k += UVOL_BLOCK) {
    const WT v = lds[k]; if (!v) continue;
    const uint32_t l = leaf0 + k / 17, w = k % 17;
    if (w == 16) atomicAdd(&V.stW[l], (unsigned long long)v);
    else {
      if (v & lo) atomicAdd(&V.stS[(size_t)l * 16 + w], (unsigned long long)(v & lo));
      if (v >> QS) atomicAdd(&V.stQ[(size_t)l * 16 + w], (unsigned long long)(v >> QS));
    }
  }
}
template <typename WT> __global__ void __launch_bounds__(UVOL_BLOCK) k_sel_stats(TexJob *job, int force, uint32_t lcap, uint32_t leaf_base) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[1];
  if ((V.done && !force) || V.nl <= leaf_base) return;
  UVOL_DYN_SMEM(WT, lds);                       // [lcap * 17]: the leaves [leaf_base, leaf_base + lcap) of this pass
  const uint32_t nl = V.nl - leaf_base, ncap = nl < lcap ? nl : lcap;
  for (uint32_t k = threadIdx.x; k < ncap * 17; k += UVOL_BLOCK) lds[k] = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, n_items = V.n_items;
  const uint32_t *const leaf = V.leaf, *const item = J.item, *const bsel = J.bsel;
  for (uint32_t base = blockIdx.x * (UVOL_BLOCK * SEL_ILP); base < n_items; base += gridDim.x * (UVOL_BLOCK * SEL_ILP)) {
    uint32_t l[SEL_ILP], it[SEL_ILP], sw[SEL_ILP];
#pragma unroll
    for (int k = 0; k < SEL_ILP; k++) { const uint32_t i = base + k * UVOL_BLOCK + threadIdx.x; const bool in = i < n_items; l[k] = in ? leaf[i] : 0xffffffffu; it[k] = in ? item[i] : 0u; }
#pragma unroll
    for (int k = 0; k < SEL_ILP; k++) sw[k] = l[k] != 0xffffffffu ? bsel[it[k]] : 0u;
#pragma unroll
    for (int k = 0; k < SEL_ILP; k++) sel_stats_accumulate(lds, l[k] != 0xffffffffu && l[k] - leaf_base < ncap, l[k] - leaf_base, sw[k], lane);   // leaves outside this pass's window: another pass
  }
  __syncthreads();
  sel_stats_flush(V, lds, ncap, leaf_base);
}
// One round of the selector tree build, after k_vq_decide<16, true>: items of a chosen leaf whose coordinate on the split axis
// exceeds the threshold move to the leaf's new sibling (what k_vq_apply does), and the moved items' statistics are accumulated
// for the new leaves [nl_old + win_base, + lcap) - the only statistics a round changes, apart from the parents' loss, which
// the next k_vq_decide subtracts.  Window 0 moves the items; a later window (more than lcap new leaves in one round) finds
// them by their new leaf.
template <typename WT> __global__ void __launch_bounds__(UVOL_BLOCK) k_sel_split_stats(TexJob *job, uint32_t lcap, uint32_t win_base) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[1];
  if (!V.round_active || V.m_round <= win_base) return;
  UVOL_DYN_SMEM(WT, lds);
  const uint32_t m = V.m_round - win_base, ncap = m < lcap ? m : lcap, nl_old = V.nl - V.m_round, leaf0 = nl_old + win_base;
  for (uint32_t k = threadIdx.x; k < ncap * 17; k += UVOL_BLOCK) lds[k] = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, n_items = V.n_items;
  uint32_t *const leaf = V.leaf; const uint32_t *const split = V.split, *const item = J.item, *const bsel = J.bsel;
  // SEL_ILP items per thread and trip: the two dependent chains of an item (leaf -> split word, item -> selector word) are in
  // flight for all of them at once; with one item per trip a workgroup spent its time waiting for four round trips in a row
  for (uint32_t base = blockIdx.x * (UVOL_BLOCK * SEL_ILP); base < n_items; base += gridDim.x * (UVOL_BLOCK * SEL_ILP)) {
    uint32_t l[SEL_ILP], it[SEL_ILP], sw[SEL_ILP], sp[SEL_ILP];
#pragma unroll
    for (int k = 0; k < SEL_ILP; k++) { const uint32_t i = base + k * UVOL_BLOCK + threadIdx.x; const bool in = i < n_items; l[k] = in ? leaf[i] : 0xffffffffu; it[k] = in ? item[i] : 0u; }
#pragma unroll
    for (int k = 0; k < SEL_ILP; k++) { sw[k] = l[k] != 0xffffffffu ? bsel[it[k]] : 0u; sp[k] = (win_base == 0 && l[k] < nl_old) ? split[l[k]] : 0u; }
#pragma unroll
    for (int k = 0; k < SEL_ILP; k++) {
      bool todo = false; uint32_t nl_ = l[k];
      if (win_base == 0) {
        if ((sp[k] >> 31) && ((sw[k] >> (2 * ((sp[k] >> 24) & 15u))) & 3u) > ((sp[k] >> 16) & 0xffu)) { nl_ = sp[k] & 0xffffu; leaf[base + k * UVOL_BLOCK + threadIdx.x] = nl_; todo = true; }
      } else todo = l[k] != 0xffffffffu && l[k] >= leaf0;
      todo = todo && nl_ - leaf0 < ncap;
      sel_stats_accumulate(lds, todo, nl_ - leaf0, sw[k], lane);
    }
  }
  __syncthreads();
  sel_stats_flush(V, lds, ncap, leaf0);
}
// split decision — one workgroup
// INCR (selector VQ): the leaf statistics are kept across rounds.  A split moves the items above the threshold to a new leaf;
// k_sel_split_stats accumulates the statistics of exactly those items into the new leaf's (zero) slots, and the next call
// here subtracts them from the parent - exact integer sums, so the result equals a recount of every leaf (which is what the
// rounds did before: one to three full passes over all items per round).  fold_only: after the last round.
template <int DIM, bool INCR>
__global__ void __launch_bounds__(UVOL_BLOCK) k_vq_decide(TexJob *job, int fold_only) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[DIM == 4 ? 0 : 1];
  if (INCR && V.round_active) {                      // uniform: written by the previous launch
    const uint32_t nlp = V.nl - V.m_round;
    for (uint32_t l = threadIdx.x; l < nlp; l += UVOL_BLOCK) {
      if (!V.chosen[l]) continue;
      const uint32_t ch = V.newidx[l];
      V.stW[l] -= V.stW[ch];
      for (int d = 0; d < DIM; d++) { V.stS[(size_t)l * DIM + d] -= V.stS[(size_t)ch * DIM + d]; V.stQ[(size_t)l * DIM + d] -= V.stQ[(size_t)ch * DIM + d]; }
    }
    __syncthreads();
  }
  if (V.done || fold_only) { __syncthreads(); if (threadIdx.x == 0) V.round_active = 0; return; }
  const uint32_t nl = V.nl, K = V.K;
  __shared__ uint32_t s_navail, s_carry;
  if (threadIdx.x == 0) { s_navail = 0; s_carry = 0; }
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t l = threadIdx.x; l < nl; l += UVOL_BLOCK) {
    const long long W = (long long)V.stW[l]; long long D = 0, best = -1; int ax = 0;
    for (int d = 0; d < DIM; d++) {
      const long long S = (long long)V.stS[(size_t)l * DIM + d], Q = (long long)V.stQ[(size_t)l * DIM + d];
      const long long num = (W * Q - S * S) * vq_wd<DIM>(d);
      D += num; if (num > best) { best = num; ax = d; }
    }
    const uint8_t sp = D > 0 ? 1 : 0;
    V.splittable[l] = sp; V.axis[l] = ax; V.th[l] = W ? (long long)V.stS[(size_t)l * DIM + ax] / W : 0; V.prio[l] = W ? D / W : 0;
    mine += sp;
  }
  if (mine) atomicAdd(&s_navail, mine);
  __syncthreads();
  const uint32_t navail = s_navail;
  if (navail == 0) { if (threadIdx.x == 0) { V.done = 1; V.round_active = 0; } return; }
  const uint32_t room = K - nl, m = navail < room ? navail : room;
  for (uint32_t l = threadIdx.x; l < nl; l += UVOL_BLOCK) {
    uint8_t ch = V.splittable[l];
    if (ch && navail > room) {
      const long long p = V.prio[l]; uint32_t rank = 0;
      for (uint32_t j = 0; j < nl; j++) if (V.splittable[j]) { const long long pj = V.prio[j]; rank += (pj > p || (pj == p && j < l)) ? 1u : 0u; }
      ch = rank < m ? 1 : 0;
    }
    V.chosen[l] = ch;
  }
  __syncthreads();
  // newidx[l] = nl + #chosen before l  (chunked block scan, every thread participates)
  for (uint32_t b0 = 0; b0 < nl; b0 += UVOL_BLOCK) {
    const uint32_t l = b0 + threadIdx.x;
    uint32_t v = l < nl ? V.chosen[l] : 0, tot;
    const uint32_t ex = t_block_excl_scan(v, &tot);
    const uint32_t c = s_carry;
    if (l < nl) { V.newidx[l] = nl + c + ex; if (INCR) V.split[l] = v ? (0x80000000u | ((uint32_t)V.axis[l] << 24) | ((uint32_t)V.th[l] << 16) | (nl + c + ex)) : 0u; }
    __syncthreads();
    if (threadIdx.x == 0) s_carry = c + tot;
    __syncthreads();
  }
  // the round's bookkeeping, formerly two more launches per round (k_vq_advance after the split, k_vq_zero before the next
  // statistics pass): k_vq_apply keys on round_active, not on done, so the leaf count can advance here; the statistics
  // have all been read above (every thread passed the barriers of the scan), so they are cleared for the next round here
  if (threadIdx.x == 0) { V.m_round = m; V.round_active = 1; V.nl = nl + m; if (nl + m >= K) V.done = 1; }
  if (!INCR) for (uint32_t i = threadIdx.x; i < K * DIM; i += UVOL_BLOCK) { if (i < K) V.stW[i] = 0; V.stS[i] = 0; V.stQ[i] = 0; }
}
template <int DIM>
__global__ void __launch_bounds__(UVOL_BLOCK) k_vq_apply(TexJob *job) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[DIM == 4 ? 0 : 1];
  if (!V.round_active) return;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (i >= V.n_items) return;
  const uint32_t l = V.leaf[i];
  if (V.chosen[l]) {
    int x[DIM]; unsigned long long w; vq_item<DIM>(J, i, x, w);
    const int ax = V.axis[l];
    int xv = 0; for (int d = 0; d < DIM; d++) xv = d == ax ? x[d] : xv;
    if ((long long)xv > V.th[l]) V.leaf[i] = V.newidx[l];
  }
}
// endpoint Lloyd: cluster centroid -> legal (colour5, inten) tuple; nearest-entry reassignment
__global__ void __launch_bounds__(UVOL_BLOCK) k_ep_entries(TexJob *job) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[0];
  const uint32_t k = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (k >= V.nl) return;
  const long long W = (long long)V.stW[k]; if (!W) return;
  int c5[3];
  for (int d = 0; d < 3; d++) { const int m8 = (int)((2 * (long long)V.stS[(size_t)k * 4 + d] + W) / (2 * W)); c5[d] = t_clampi((m8 * 31 + 127) / 255, 0, 31); }
  const int mi = (int)((2 * (long long)V.stS[(size_t)k * 4 + 3] + W) / (2 * W));
  int bt = 0, bd = 1 << 30;
  for (int t = 0; t < 8; t++) { int d = t_inten(t, 3) - mi; d = d < 0 ? -d : d; if (d < bd) { bd = d; bt = t; } }
  J.ent[k] = ((uint32_t)bt << 15) | ((uint32_t)c5[0] << 10) | ((uint32_t)c5[1] << 5) | (uint32_t)c5[2];
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_ep_assign(TexJob *job) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[0];
  __shared__ int se[UVOL_BLOCK * 4];
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const bool live = i < V.n_items;
  int x[4] = {0, 0, 0, 0}; if (live) t_cell_coords(J.cid[i], x);
  int bd = 0x7fffffff; uint32_t bk = 0;
  for (uint32_t k0 = 0; k0 < V.nl; k0 += UVOL_BLOCK) {
    __syncthreads();
    if (k0 + threadIdx.x < V.nl) { int e[4]; t_cell_coords(J.ent[k0 + threadIdx.x], e); for (int d = 0; d < 4; d++) se[threadIdx.x * 4 + d] = e[d]; }
    __syncthreads();
    const uint32_t kn = V.nl - k0 < UVOL_BLOCK ? V.nl - k0 : UVOL_BLOCK;
    if (live) for (uint32_t k = 0; k < kn; k++) {
      const int d0 = x[0] - se[k * 4], d1 = x[1] - se[k * 4 + 1], d2 = x[2] - se[k * 4 + 2], d3 = x[3] - se[k * 4 + 3];
      const int dist = d0 * d0 + d1 * d1 + d2 * d2 + 2 * d3 * d3;
      if (dist < bd) { bd = dist; bk = k0 + k; }
    }
  }
  if (live) V.leaf[i] = bk;
}

// unique + ascending order of the used cluster values (single workgroup, two O(K^2) passes).
// which = 0: endpoints (values J.ent, used = stW>0) -> J.ecb / J.emap / J.ne ; 1: selectors (J.scb, J.sused) -> J.scu / J.smap / J.ns
__global__ void __launch_bounds__(UVOL_BLOCK) k_unique(TexJob *job, int which) {
  TJOB_OR_RETURN;
  const uint32_t K = J.vq[which].nl;
  const uint32_t *val = which == 0 ? J.ent : J.scb;
  uint32_t *outv = which == 0 ? J.ecb : J.scu, *map = which == 0 ? J.emap : J.smap;
  uint8_t *first = J.vq[which].splittable;       // scratch: 1 = used and first occurrence of its value, 2 = used duplicate, 0 = unused
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < K; k += UVOL_BLOCK) {
    const bool used = which == 0 ? (J.vq[0].stW[k] != 0) : (J.sused[k] != 0);
    uint8_t f = 0;
    if (used) { f = 1; const uint32_t v = val[k]; for (uint32_t j = 0; j < k; j++) { const bool uj = which == 0 ? (J.vq[0].stW[j] != 0) : (J.sused[j] != 0); if (uj && val[j] == v) { f = 2; break; } } }
    first[k] = f;
  }
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t k = threadIdx.x; k < K; k += UVOL_BLOCK) {
    if (!first[k]) { map[k] = 0; continue; }
    const uint32_t v = val[k]; uint32_t less = 0;
    for (uint32_t j = 0; j < K; j++) less += (first[j] == 1 && val[j] < v) ? 1u : 0u;
    map[k] = less;
    if (first[k] == 1) { outv[less] = v; mine++; }
  }
  if (mine) atomicAdd(&s_cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) { if (which == 0) J.ne = s_cnt; else J.ns = s_cnt; }
}

// per coded block: endpoint index, optimal selectors under the codebook endpoint, "coded" flag for the item list
__global__ void __launch_bounds__(UVOL_BLOCK) k_block_assign(TexJob *job) {
  TJOB_OR_RETURN;
  const uint32_t it = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (it >= J.n_items) return;
  const uint32_t b = J.item[it];
  const uint32_t cl = J.vq[0].leaf[J.cidx[J.cell[b]]], tup = J.ent[cl];
  J.bei[b] = (uint16_t)J.emap[cl];
  const uint32_t l = b / J.nb, r = b % J.nb;
  uint32_t px[16]; t_load_block(J, l, r % J.bx, r / J.bx, px);
  uint32_t sel;
  t_eval_block<true, false>(px, (int)((tup >> 10) & 31), (int)((tup >> 5) & 31), (int)(tup & 31), (int)(tup >> 15), &sel, nullptr);
  J.bsel[b] = sel;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_item_compact(TexJob *job) {
  TexJob &J = job[blockIdx.z];
  const uint32_t b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const bool live = J.status == 0 && b < J.NB;
  uint32_t v = live ? J.flag[b] : 0, tot;
  const uint32_t pos = t_block_excl_scan(v, &tot) + J.bsum[blockIdx.x];
  if (live && v) { J.item[pos] = b; J.vq[1].leaf[pos] = 0; }
  if (blockIdx.x == 0 && threadIdx.x == 0 && J.status == 0) {
    const uint32_t n = J.bsum[(J.NB + UVOL_BLOCK - 1) / UVOL_BLOCK];
    J.n_items = n;
    TexVQ &V = J.vq[1]; V.n_items = n; V.K = n < J.Kmax_s ? n : J.Kmax_s; V.nl = 1; V.done = (V.K <= 1) ? 1 : 0;
  }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_cell_leaf_init(TexJob *job) {
  TJOB_OR_RETURN;
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (i < J.vq[0].n_items) J.vq[0].leaf[i] = 0;
}

// selector Lloyd
__global__ void __launch_bounds__(UVOL_BLOCK) k_sel_centroids(TexJob *job) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[1];
  const uint32_t k = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (k >= V.nl) return;
  const long long W = (long long)V.stW[k]; if (!W) return;
  uint32_t v = 0;
  for (int d = 0; d < 16; d++) v |= (uint32_t)((2 * (long long)V.stS[(size_t)k * 16 + d] + W) / (2 * W)) << (2 * d);
  J.scb[k] = v;
}
// assignment by true block SSE.  The SSE of item j under codebook entry i is sum_t E_j[t][sel_i(t)] = a 64-long dot product of
// the item's per-texel error table (16 texels x 4 selector values, u16 each) with the entry's one-hot selector vector: a
// nearest-codeword search, which is what the matrix cores are for.  The u16 errors go in as two signed-i8 planes (byte ^ 0x80,
// i.e. byte - 128; every one-hot vector has exactly 16 ones, so starting the accumulator at 16 * 128 restores the true sum),
// one v_mfma_i32_16x16x64_i8 per plane per (16 entries x 16 items) tile, exact in the i32 accumulators.  k = 4 * texel +
// selector, lane (x, g) holds k-group g = texels 4g..4g+3 for both operands.  Each wave keeps 4 item tiles (64 items) in
// registers and streams the codebook (one-hot rows in LDS, chunks of SA_CHUNK entries) past them; the running minimum is one
// v_min_u32 on (sse << 11 | entry), which also gives the lowest entry among equal SSEs like the ascending scan it replaces.
// Rows past nl repeat entry nl - 1 (same SSE, larger index: never selected).
#define SA_CHUNK 512u
__global__ void __launch_bounds__(UVOL_BLOCK) k_sel_assign(TexJob *job) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[1];
  if (blockIdx.x * UVOL_BLOCK >= V.n_items) return;
  __shared__ uint32_t lds[SA_CHUNK * 16];                              // 32 KiB: item planes first, then code chunks
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, i = blockIdx.x * UVOL_BLOCK + tid;
  {
    unsigned long long rows[16];
    for (int t = 0; t < 16; t++) rows[t] = 0;
    if (i < V.n_items) {
      const uint32_t b = J.item[i], tup = J.ecb[J.bei[b]];
      const uint32_t l = b / J.nb, r = b % J.nb;
      uint32_t px[16]; t_load_block(J, l, r % J.bx, r / J.bx, px);
      t_eval_block<false, true>(px, (int)((tup >> 10) & 31), (int)((tup >> 5) & 31), (int)(tup & 31), (int)(tup >> 15), nullptr, rows);
    }
    // [wave][tile][plane][g][x][w]: what lane (x, g) reads back as one 16-byte fragment
    const uint32_t tile = lane >> 4, x = lane & 15u;
    for (int t = 0; t < 16; t++) {
      const uint32_t lo32 = (uint32_t)rows[t], hi32 = (uint32_t)(rows[t] >> 32);
      const uint32_t lo = (lo32 & 0xffu) | ((lo32 >> 8) & 0xff00u) | ((hi32 & 0xffu) << 16) | ((hi32 << 8) & 0xff000000u);
      const uint32_t hi = ((lo32 >> 8) & 0xffu) | ((lo32 >> 16) & 0xff00u) | ((hi32 << 8) & 0xff0000u) | (hi32 & 0xff000000u);
      const uint32_t g = (uint32_t)t >> 2, w = (uint32_t)t & 3u;
      lds[((((wv * 4 + tile) * 2 + 0) * 4 + g) * 16 + x) * 4 + w] = lo ^ 0x80808080u;
      lds[((((wv * 4 + tile) * 2 + 1) * 4 + g) * 16 + x) * 4 + w] = hi ^ 0x80808080u;
    }
  }
  __syncthreads();
  uint32_t blo[4][4], bhi[4][4], best[4];
  for (uint32_t tile = 0; tile < 4; tile++) {
    const uint32_t o0 = (((wv * 4 + tile) * 2 + 0) * 64 + lane) * 4, o1 = (((wv * 4 + tile) * 2 + 1) * 64 + lane) * 4;
    for (int w = 0; w < 4; w++) { blo[tile][w] = lds[o0 + w]; bhi[tile][w] = lds[o1 + w]; }
    best[tile] = 0xffffffffu;
  }
  const uint32_t nl = V.nl;
  for (uint32_t c0 = 0; c0 < nl; c0 += SA_CHUNK) {
    const uint32_t cn = nl - c0 < SA_CHUNK ? nl - c0 : SA_CHUNK, ctn = (cn + 15u) >> 4;
    __syncthreads();
    // one-hot rows, [code tile][g][x][w]: word (g, w) of entry x = 1 << 8 * selector(texel 4g + w)
    for (uint32_t idx = tid; idx < ctn * 256u; idx += UVOL_BLOCK) {
      const uint32_t ct = idx >> 8, rem = idx & 255u, g = rem >> 6, x = (rem >> 2) & 15u, w = rem & 3u;
      uint32_t k = c0 + ct * 16 + x; if (k >= nl) k = nl - 1;
      lds[idx] = 1u << (8u * ((J.scb[k] >> (2u * (4u * g + w))) & 3u));
    }
    __syncthreads();
    uint32_t kbr[4];                                                   // entry index of accumulator row r, kept as four registers
    for (uint32_t r = 0; r < 4; r++) { kbr[r] = c0 + 4 * (lane >> 4) + r; UVOL_OPAQUE(kbr[r]); }
    for (uint32_t ct = 0; ct < ctn; ct++) {
      uint32_t a[4];
      for (int w = 0; w < 4; w++) a[w] = lds[(ct * 64 + lane) * 4 + w];
#pragma unroll
      for (int tile = 0; tile < 4; tile++) {
        int H[4] = { 2048, 2048, 2048, 2048 }, L[4] = { 2048, 2048, 2048, 2048 };
        t_mfma_i8_16x16x64(a, bhi[tile], H);
        t_mfma_i8_16x16x64(a, blo[tile], L);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          uint32_t d = ((uint32_t)H[r] << 8) + (uint32_t)L[r];
          UVOL_OPAQUE(d);                                              // keep (H << 8) + L one v_lshl_add, then one v_lshl_or
          const uint32_t key = (d << 11) | kbr[r];
          best[tile] = key < best[tile] ? key : best[tile];
        }
      }
      for (uint32_t r = 0; r < 4; r++) kbr[r] += 16;
    }
  }
  for (uint32_t tile = 0; tile < 4; tile++) {
    uint32_t m = best[tile], o;
    o = __shfl_xor(m, 16); m = o < m ? o : m;
    o = __shfl_xor(m, 32); m = o < m ? o : m;
    const uint32_t it = blockIdx.x * UVOL_BLOCK + wv * 64 + tile * 16 + (lane & 15u);
    if (lane < 16 && it < V.n_items) V.leaf[it] = m & 2047u;
  }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_sel_used(TexJob *job, int phase) {
  TJOB_OR_RETURN;
  TexVQ &V = J.vq[1];
  const uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (phase == 0) { if (i < V.K) J.sused[i] = 0; }
  else if (phase == 1) { if (i < V.n_items) J.sused[V.leaf[i]] = 1; }
  else { if (i < V.n_items) J.bsi[J.item[i]] = (uint16_t)J.smap[V.leaf[i]]; }
}
// skipped blocks copy the previous layer's final indices (sequential in the layer index)
__global__ void __launch_bounds__(UVOL_BLOCK) k_copy_skipped(TexJob *job) {
  TJOB_OR_RETURN;
  const uint32_t b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (b >= J.nb) return;
  uint32_t nsk = 0;
  const uint32_t st = 1u << J.ashift;                     // the previous slice of the same kind
  for (uint32_t l = st; l < J.L; l++) {
    const size_t o = (size_t)l * J.nb + b, pv = o - (size_t)st * J.nb;
    if (J.skip[o]) { J.bei[o] = J.bei[pv]; J.bsi[o] = J.bsi[pv]; nsk++; }
  }
  if (nsk) atomicAdd(&J.n_skipped, nsk);
}

// ------------------------------------------------------------------------------------------------
// K12: BasisLZ slices.  preds (parallel) -> symbolisation (one wave per slice, lanes hold the 64-entry
// selector history) -> Huffman tables -> prefix-scan of code lengths -> atomicOr bit packing.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t t_pred_of(const TexJob &J, uint32_t l, uint32_t x, uint32_t y) {
  const size_t o = (size_t)l * J.nb; const uint32_t b = y * J.bx + x;
  const uint16_t *ei = J.bei + o;
  if (J.skip[o + b]) return 2;                            // (never set in the first slice of a kind)
  if (x > 0 && ei[b] == ei[b - 1]) return 0;
  if (y > 0 && ei[b] == ei[b - J.bx]) return 1;
  if ((l >> J.ashift) == 0 && x > 0 && y > 0 && ei[b] == ei[b - J.bx - 1]) return 2;
  return 3;
}
// pred[b] bits 0-1: this block's predictor; for macroblock-origin blocks bits 8.. are not stored: the
// macro symbol is rebuilt from the four preds by the symboliser
__global__ void __launch_bounds__(UVOL_BLOCK) k_preds(TexJob *job) {
  TJOB_OR_RETURN;
  const uint32_t b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (b >= J.NB) return;
  const uint32_t l = b / J.nb, r = b % J.nb;
  J.pred[b] = (uint8_t)t_pred_of(J, l, r % J.bx, r / J.bx);
}

#define TOK(kind, sym, extra) ((unsigned long long)(kind) | ((unsigned long long)(sym) << 8) | ((unsigned long long)(extra) << 32))
#define TOK_NOP 255ull

// ---- per-slice scans: flags at J.flag[l*stride + i], block sums at J.bsum[l*(nblk+1) + blk] ----
__global__ void __launch_bounds__(UVOL_BLOCK) k_sscan_a(TexJob *job, uint32_t n, uint32_t stride) {
  TexJob &J = job[blockIdx.z];
  const uint32_t l = blockIdx.y, i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  uint32_t v = (J.status == 0 && i < n) ? J.flag[(size_t)l * stride + i] : 0, tot;
  t_block_excl_scan(v, &tot);
  if (threadIdx.x == 0) J.bsum[(size_t)l * (gridDim.x + 1) + blockIdx.x] = tot;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_sscan_b(TexJob *job, uint32_t nblocks) {
  TexJob &J = job[blockIdx.z];
  uint32_t *bs = J.bsum + (size_t)blockIdx.x * (nblocks + 1);
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nblocks; b0 += UVOL_BLOCK) {
    const uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < nblocks ? bs[i] : 0, tot;
    const uint32_t ex = t_block_excl_scan(v, &tot);
    const uint32_t c = carry;
    if (i < nblocks) bs[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) bs[nblocks] = carry;
}

// delta-endpoint tokens + slot initialisation + "coded block" flags (parallel; prev_ei is simply the raster predecessor)
__global__ void __launch_bounds__(UVOL_BLOCK) k_tok_delta(TexJob *job) {
  TJOB_OR_RETURN;
  const uint32_t b = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (b >= J.NB) return;
  const uint32_t l = b / J.nb, r = b % J.nb;
  const uint32_t e = J.bei[b], pe = r ? J.bei[b - 1] : 0u;
  unsigned long long *T = J.tok + 3 * (size_t)b;
  T[0] = TOK_NOP; T[2] = TOK_NOP;
  T[1] = J.pred[b] == 3 ? TOK(2, e >= pe ? e - pe : e + J.ne - pe, 0) : TOK_NOP;
  J.flag[b] = J.skip[b] ? 0 : 1;
}
// coded blocks of each slice, in raster order
__global__ void __launch_bounds__(UVOL_BLOCK) k_coded_list(TexJob *job) {
  TexJob &J = job[blockIdx.z];
  const uint32_t l = blockIdx.y, i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const bool live = J.status == 0 && i < J.nb;
  uint32_t v = live ? J.flag[(size_t)l * J.nb + i] : 0, tot;
  const uint32_t pos = t_block_excl_scan(v, &tot) + J.bsum[(size_t)l * (gridDim.x + 1) + blockIdx.x];
  if (live && v) J.clist[(size_t)l * J.nb + pos] = i;
  if (blockIdx.x == 0 && threadIdx.x == 0) J.ncoded[l] = J.status == 0 ? J.bsum[(size_t)l * (gridDim.x + 1) + gridDim.x] : 0;
}
// macroblock symbols (2x2 blocks, 2 bits each) and equal-value group boundaries
__device__ __forceinline__ uint32_t t_macro_sym(const TexJob &J, uint32_t l, uint32_t k) {
  const uint32_t mbx = (J.bx + 1) / 2, x = 2 * (k % mbx), y = 2 * (k / mbx);
  const uint8_t *pr = J.pred + (size_t)l * J.nb; const uint32_t b = y * J.bx + x;
  uint32_t ms = pr[b];
  if (x + 1 < J.bx) ms |= (uint32_t)pr[b + 1] << 2;
  if (y + 1 < J.by) { ms |= (uint32_t)pr[b + J.bx] << 4; if (x + 1 < J.bx) ms |= (uint32_t)pr[b + J.bx + 1] << 6; }
  return ms;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_mb_flags(TexJob *job, uint32_t nm) {
  TJOB_OR_RETURN;
  const uint32_t l = blockIdx.y, k = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (k >= nm) return;
  const uint32_t ms = t_macro_sym(J, l, k), pm = k ? t_macro_sym(J, l, k - 1) : 0u;
  J.msym[(size_t)l * nm + k] = (uint8_t)ms;
  J.flag[(size_t)l * nm + k] = ms != pm ? 1 : 0;          // group boundary (the decoder's prev_sym starts at 0)
  J.gsize[(size_t)l * nm + k] = 0;
}
// gid = number of boundaries up to and including k (0 = the leading group that continues prev_sym = 0)
__global__ void __launch_bounds__(UVOL_BLOCK) k_mb_groups(TexJob *job, uint32_t nm) {
  TexJob &J = job[blockIdx.z];
  const uint32_t l = blockIdx.y, k = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  const bool live = J.status == 0 && k < nm;
  uint32_t v = live ? J.flag[(size_t)l * nm + k] : 0, tot;
  const uint32_t ex = t_block_excl_scan(v, &tot) + J.bsum[(size_t)l * (gridDim.x + 1) + blockIdx.x];
  if (!live) return;
  const uint32_t g = ex + v;
  J.gid[(size_t)l * nm + k] = g;
  if (v || k == 0) J.gstart[(size_t)l * nm + g] = k;
  atomicAdd(&J.gsize[(size_t)l * nm + g], 1u);
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_mb_tokens(TexJob *job, uint32_t nm) {
  TJOB_OR_RETURN;
  const uint32_t l = blockIdx.y, k = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  if (k >= nm) return;
  const size_t o = (size_t)l * nm;
  const uint32_t g = J.gid[o + k], st = J.gstart[o + g], sz = J.gsize[o + g], ms = J.msym[o + k];
  const uint32_t lit = g > 0 ? 1u : 0u;                    // groups > 0 open with a literal symbol
  const uint32_t run_first = st + lit, run_len = sz - lit;
  const uint32_t mbx = (J.bx + 1) / 2, b = (2 * (k / mbx)) * J.bx + 2 * (k % mbx);
  unsigned long long tk = TOK_NOP;
  if (lit && k == st) tk = TOK(0, ms, 0);
  else if (run_len >= 3) { if (k == run_first) tk = TOK(1, 256, run_len - 3); }
  else tk = TOK(0, ms, 0);
  J.tok[3 * ((size_t)l * J.nb + b)] = tk;
}

// selector tokens: the 64-entry history (move-towards-front by index halving, rover insertion) is inherently
// sequential.  One wave per slice; lane k keeps history entry k in a register (search = ballot + ffs, swap = two
// lane reads); the coded blocks are prefetched 64 at a time; tokens are collected per lane and stored coalesced.
__global__ void __launch_bounds__(64) k_sel_tokens(TexJob *job) {
  TexJob &J = job[blockIdx.z];
  UVOL_SERIAL_PRIO();
  const uint32_t l = blockIdx.x, lane = threadIdx.x;
  const bool ok = J.status == 0;
  const uint32_t n = ok ? J.ncoded[l] : 0, ns = J.ns;
  const size_t o = (size_t)l * J.nb;
  const uint32_t *cl = J.clist + o; const uint16_t *si = J.bsi + o;
  unsigned long long *T = J.tok + 3 * o;
  uint32_t hist = lane, rover = TEX_HS / 2;
  uint32_t run = 0, run_e1 = 0, run_e2 = 0;            // pending history[0] run: length and element indices of its first two members
  auto load_chunk = [&](uint32_t base, uint32_t &B, uint32_t &S) { const uint32_t e = base + lane; B = 0; S = 0; if (e < n) { B = cl[e]; S = si[B]; } };
  uint32_t nB = 0, nS = 0;
  if (n) load_chunk(0, nB, nS);
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t cB = nB, cS = nS;
    if (base + 64 < n) load_chunk(base + 64, nB, nS);
    unsigned long long mytok = TOK_NOP;
    const uint32_t cnt = n - base < 64 ? n - base : 64;
    // finalise a pending run; its members may sit in this chunk (lane registers) or in an earlier one (direct store)
#define PUT_TOK(e, tk) do { if ((e) >= base) { if (lane == (e) - base) mytok = (tk); } else if (lane == 0) T[3 * (size_t)cl[(e)] + 2] = (tk); } while (0)
#define FIN_RUN() do { if (run >= 3) { const uint32_t rs_ = run - 3 < 63 ? run - 3 : 63; PUT_TOK(run_e1, TOK(4, rs_, rs_ == 63 ? run - 3 : 0)); } \
      else { if (run >= 1) PUT_TOK(run_e1, TOK(3, ns, 0)); if (run == 2) PUT_TOK(run_e2, TOK(3, ns, 0)); } run = 0; } while (0)
    for (uint32_t j = 0; j < cnt; j++) {
      const uint32_t s = UVOL_READLANE(cS, j);
      const unsigned long long hit = __ballot(hist == s);
      if (hit & 1ull) { run++; if (run == 1) run_e1 = base + j; else if (run == 2) run_e2 = base + j; continue; }
      FIN_RUN();
      if (hit) {
        const uint32_t h = (uint32_t)(__ffsll((long long)hit) - 1);
        if (lane == j) mytok = TOK(3, ns + h, 0);
        const uint32_t va = UVOL_READLANE(hist, h), vc = UVOL_READLANE(hist, h / 2);
        if (lane == h) hist = vc; else if (lane == h / 2) hist = va;
      } else {
        if (lane == j) mytok = TOK(3, s, 0);
        if (lane == rover) hist = s;
        rover++; if (rover == TEX_HS) rover = TEX_HS / 2;
      }
    }
    if (base + 64 >= n) FIN_RUN();
    if (lane < cnt) T[3 * (size_t)cB + 2] = mytok;
    // a run that stays open across the chunk boundary keeps NOPs in this chunk; its head is patched by a direct store later
  }
#undef PUT_TOK
#undef FIN_RUN
}

// symbol histograms of the four slice models from the finished token slots (LDS-privatised)
__global__ void __launch_bounds__(UVOL_BLOCK) k_tok_hist(TexJob *job) {
  TJOB_OR_RETURN;
  UVOL_DYN_SMEM(uint32_t, lh);
  const uint32_t o_de = 257, o_sel = 257 + J.Kmax_e, o_rle = o_sel + J.Kmax_s + TEX_HS + 1, n_lh = o_rle + 64;
  for (uint32_t k = threadIdx.x; k < n_lh; k += UVOL_BLOCK) lh[k] = 0;
  __syncthreads();
  const size_t n = 3 * (size_t)J.NB;
  for (size_t i = (size_t)blockIdx.x * UVOL_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * UVOL_BLOCK) {
    const unsigned long long tk = J.tok[i];
    const uint32_t kind = (uint32_t)(tk & 255), sym = (uint32_t)((tk >> 8) & 0xffffff);
    if (kind == 255) continue;
    if (kind <= 1) atomicAdd(&lh[kind == 1 ? 256u : sym], 1u);
    else if (kind == 2) atomicAdd(&lh[o_de + sym], 1u);
    else if (kind == 3) atomicAdd(&lh[o_sel + sym], 1u);
    else { atomicAdd(&lh[o_sel + J.ns + TEX_HS], 1u); atomicAdd(&lh[o_rle + sym], 1u); }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n_lh; k += UVOL_BLOCK) {
    const uint32_t v = lh[k]; if (!v) continue;
    if (k < o_de) atomicAdd(&J.hm[0].freq[k], v); else if (k < o_sel) atomicAdd(&J.hm[1].freq[k - o_de], v);
    else if (k < o_rle) atomicAdd(&J.hm[2].freq[k - o_sel], v); else atomicAdd(&J.hm[3].freq[k - o_rle], v);
  }
}

// ---- cooperative (one workgroup) Huffman construction; mirrors the CPU restatement exactly ----
// scratch layout (uint32): sym[n] | f[n] | parent[2n] | w_lo[2n] w_hi[2n]
__device__ inline void dev_huff_build(const uint32_t *freq_in, int n, int maxlen, uint8_t *size, uint16_t *code, uint32_t *scratch) {
  __shared__ int s_m;
  uint32_t *sym = scratch, *fs = scratch + n + 1; int *parent = (int *)(scratch + 2 * (n + 1));
  unsigned long long *w = (unsigned long long *)(((uintptr_t)(scratch + 4 * (n + 1) + 2) + 7) & ~(uintptr_t)7);
  if (threadIdx.x == 0) s_m = 0;
  for (int i = threadIdx.x; i < n; i += UVOL_BLOCK) { size[i] = 0; code[i] = 0; }
  __syncthreads();
  // stable rank among used symbols by (freq, sym)
  for (int i = threadIdx.x; i < n; i += UVOL_BLOCK) {
    const uint32_t f = freq_in[i]; if (!f) continue;
    uint32_t r = 0; for (int j = 0; j < n; j++) { const uint32_t fj = freq_in[j]; if (fj && (fj < f || (fj == f && j < i))) r++; }
    sym[r] = (uint32_t)i; fs[r] = f; atomicAdd(&s_m, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int m = s_m;
    if (m == 0) { sym[0] = 0; fs[0] = 1; m = 1; }
    if (m == 1) size[sym[0]] = 1;
    else {
      for (int i = 0; i < m; i++) w[i] = fs[i];
      int li = 0, ni = m, nn = m;
      for (int k = 0; k < m - 1; k++) {
        int a, b;
        if (li < m && (ni >= nn || w[li] <= w[ni])) a = li++; else a = ni++;
        if (li < m && (ni >= nn || w[li] <= w[ni])) b = li++; else b = ni++;
        w[nn] = w[a] + w[b]; parent[a] = nn; parent[b] = nn; nn++;
      }
      parent[nn - 1] = -1;
      int cnt[64]; for (int i = 0; i < 64; i++) cnt[i] = 0;
      for (int i = 0; i < m; i++) { int d = 0, p = i; while (parent[p] >= 0) { p = parent[p]; d++; } if (d > 63) d = 63; cnt[d]++; }
      for (int l = maxlen + 1; l < 64; l++) { cnt[maxlen] += cnt[l]; cnt[l] = 0; }
      unsigned long long total = 0; for (int l = maxlen; l > 0; l--) total += (unsigned long long)cnt[l] << (maxlen - l);
      while (total != (1ull << maxlen)) {
        cnt[maxlen]--;
        for (int l = maxlen - 1; l > 0; l--) if (cnt[l]) { cnt[l]--; cnt[l + 1] += 2; break; }
        total--;
      }
      int j = 0; for (int l = maxlen; l >= 1; l--) for (int c = 0; c < cnt[l]; c++) size[sym[j++]] = (uint8_t)l;
    }
    uint32_t blc[20]; for (int i = 0; i < 20; i++) blc[i] = 0;
    for (int i = 0; i < n; i++) if (size[i]) blc[size[i]]++;
    uint32_t next[20]; uint32_t c0 = 0; next[0] = 0;
    for (int l = 1; l <= 16; l++) { c0 = (c0 + blc[l - 1]) << 1; next[l] = c0; }
    for (int i = 0; i < n; i++) if (size[i]) {
      const uint32_t c = next[size[i]]++; uint32_t r = 0; for (int k = 0; k < size[i]; k++) r |= ((c >> k) & 1) << (size[i] - 1 - k);
      code[i] = (uint16_t)r;
    }
  }
  __syncthreads();
}
// serialise a code-length table (inverse of SURVEY B.1 read_huff); thread 0 writes, all threads build
__device__ inline void dev_write_huff(TBitW &w, const uint8_t *size, int n, uint32_t *scratch) {
  __shared__ uint32_t s_f[21]; __shared__ uint8_t s_size[21]; __shared__ uint16_t s_code[21]; __shared__ int s_total, s_nt;
  uint8_t *tok = (uint8_t *)(scratch + 8 * (TEX_MODEL_CAP + 8)), *ext = tok + TEX_MODEL_CAP + 8;
  const int ZZ[21] = { 17, 18, 19, 20, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15, 16 };
  if (threadIdx.x == 0) {
    int total = 0; for (int i = 0; i < n; i++) if (size[i]) total = i + 1;
    s_total = total; tb_put(w, (uint32_t)total, 14);
    int nt = 0;
    for (int i = 0; i < 21; i++) s_f[i] = 0;
    for (int i = 0; i < total;) {
      const int len = size[i]; int run = 1; while (i + run < total && size[i + run] == len) run++;
      i += run;
      if (len == 0) {
        while (run > 0) {
          if (run < 3) { tok[nt] = 0; ext[nt++] = 0; run--; }
          else if (run <= 10) { tok[nt] = 17; ext[nt++] = (uint8_t)(run - 3); run = 0; }
          else { const int r = run > 138 ? 138 : run; tok[nt] = 18; ext[nt++] = (uint8_t)(r - 11); run -= r; }
        }
      } else {
        tok[nt] = (uint8_t)len; ext[nt++] = 0; run--;
        while (run > 0) {
          if (run < 3) { tok[nt] = (uint8_t)len; ext[nt++] = 0; run--; }
          else if (run <= 6) { tok[nt] = 19; ext[nt++] = (uint8_t)(run - 3); run = 0; }
          else { const int r = run > 134 ? 134 : run; tok[nt] = 20; ext[nt++] = (uint8_t)(r - 7); run -= r; }
        }
      }
    }
    for (int i = 0; i < nt; i++) s_f[tok[i]]++;
    s_nt = nt;
  }
  __syncthreads();
  if (s_total == 0) return;
  dev_huff_build(s_f, 21, 7, s_size, s_code, scratch);
  if (threadIdx.x == 0) {
    int ncl = 1; for (int i = 0; i < 21; i++) if (s_size[ZZ[i]]) ncl = i + 1;
    tb_put(w, (uint32_t)ncl, 5);
    for (int i = 0; i < ncl; i++) tb_put(w, s_size[ZZ[i]], 3);
    for (int i = 0; i < s_nt; i++) {
      const int t = tok[i];
      tb_put(w, s_code[t], s_size[t]);
      if (t == 17) tb_put(w, ext[i], 3); else if (t == 18) tb_put(w, ext[i], 7); else if (t == 19) tb_put(w, ext[i], 2); else if (t == 20) tb_put(w, ext[i], 7);
    }
  }
  __syncthreads();
}

// the four slice models (blockIdx.x = model id 0..3)
__global__ void __launch_bounds__(UVOL_BLOCK) k_huff_models(TexJob *job) {
  TJOB_OR_RETURN;
  const int mi = blockIdx.x;
  TexHuff &H = J.hm[mi];
  const uint32_t n = mi == 0 ? 257u : (mi == 1 ? J.ne : (mi == 2 ? J.ns + TEX_HS + 1 : 64u));
  if (threadIdx.x == 0) {
    H.n = n;
    if (mi == 1 || mi == 3) { uint32_t s = 0; for (uint32_t i = 0; i < n; i++) s |= H.freq[i]; if (!s) H.freq[0] = 1; }   // the transcoder rejects empty models
  }
  __syncthreads();
  dev_huff_build(H.freq, (int)n, 16, H.size, H.code, J.hscratch + (size_t)mi * 10 * (TEX_MODEL_CAP + 8));
}

// codebook + table sections (blockIdx.x: 0 endpoints, 1 selectors, 2 tables)
__global__ void __launch_bounds__(UVOL_BLOCK) k_sections(TexJob *job) {
  TJOB_OR_RETURN;
  const int sec = blockIdx.x;
  uint32_t *scratch = J.hscratch + (size_t)(4 + sec) * 10 * (TEX_MODEL_CAP + 8);
  __shared__ TBitW w;
  if (threadIdx.x == 0) tb_init(w, J.sec[sec], J.sec_cap[sec]);
  __syncthreads();
  if (sec == 0) {
    const uint32_t ne = J.ne;
    if (threadIdx.x == 0) {
      for (int m = 4; m <= 7; m++) { const uint32_t n = m == 7 ? 8 : 32; J.hm[m].n = n; for (uint32_t i = 0; i < n; i++) J.hm[m].freq[i] = 0; }
      int prev[3] = { 16, 16, 16 }, pi = 0;
      for (uint32_t k = 0; k < ne; k++) {
        const uint32_t e = J.ecb[k]; const int t = (int)(e >> 15), c[3] = { (int)((e >> 10) & 31), (int)((e >> 5) & 31), (int)(e & 31) };
        J.hm[7].freq[(t - pi) & 7]++; pi = t;
        for (int d = 0; d < 3; d++) { const int m = prev[d] <= 9 ? 0 : (prev[d] <= 21 ? 1 : 2); J.hm[4 + m].freq[(c[d] - prev[d]) & 31]++; prev[d] = c[d]; }
      }
      for (int m = 4; m <= 6; m++) { uint32_t s = 0; for (int i = 0; i < 32; i++) s |= J.hm[m].freq[i]; if (!s) J.hm[m].freq[0] = 1; }
    }
    __syncthreads();
    for (int m = 4; m <= 7; m++) dev_huff_build(J.hm[m].freq, (int)J.hm[m].n, 16, J.hm[m].size, J.hm[m].code, scratch);
    for (int m = 4; m <= 7; m++) dev_write_huff(w, J.hm[m].size, (int)J.hm[m].n, scratch);
    if (threadIdx.x == 0) {
      tb_put(w, 0, 1);
      int prev[3] = { 16, 16, 16 }, pi = 0;
      for (uint32_t k = 0; k < ne; k++) {
        const uint32_t e = J.ecb[k]; const int t = (int)(e >> 15), c[3] = { (int)((e >> 10) & 31), (int)((e >> 5) & 31), (int)(e & 31) };
        const uint32_t si_ = (uint32_t)((t - pi) & 7); tb_put(w, J.hm[7].code[si_], J.hm[7].size[si_]); pi = t;
        for (int d = 0; d < 3; d++) { const int m = prev[d] <= 9 ? 0 : (prev[d] <= 21 ? 1 : 2); const uint32_t s = (uint32_t)((c[d] - prev[d]) & 31); tb_put(w, J.hm[4 + m].code[s], J.hm[4 + m].size[s]); prev[d] = c[d]; }
      }
    }
  } else if (sec == 1) {
    const uint32_t ns = J.ns;
    if (threadIdx.x == 0) {
      tb_put(w, 0, 1); tb_put(w, 0, 1); tb_put(w, 0, 1);
      J.hm[8].n = 256; for (int i = 0; i < 256; i++) J.hm[8].freq[i] = 0;
      for (uint32_t k = 1; k < ns; k++) for (int j = 0; j < 4; j++) J.hm[8].freq[((J.scu[k] >> (8 * j)) ^ (J.scu[k - 1] >> (8 * j))) & 255]++;
      uint32_t s = 0; for (int i = 0; i < 256; i++) s |= J.hm[8].freq[i]; if (!s) J.hm[8].freq[0] = 1;
    }
    __syncthreads();
    dev_huff_build(J.hm[8].freq, 256, 16, J.hm[8].size, J.hm[8].code, scratch);
    dev_write_huff(w, J.hm[8].size, 256, scratch);
    if (threadIdx.x == 0) {
      for (int j = 0; j < 4; j++) tb_put(w, (J.scu[0] >> (8 * j)) & 255, 8);
      for (uint32_t k = 1; k < ns; k++) for (int j = 0; j < 4; j++) { const uint32_t s = ((J.scu[k] >> (8 * j)) ^ (J.scu[k - 1] >> (8 * j))) & 255; tb_put(w, J.hm[8].code[s], J.hm[8].size[s]); }
    }
  } else {
    for (int m = 0; m < 4; m++) dev_write_huff(w, J.hm[m].size, (int)J.hm[m].n, scratch);
    if (threadIdx.x == 0) tb_put(w, TEX_HS, 13);
  }
  __syncthreads();
  if (threadIdx.x == 0) { tb_flush(w); J.sec_len[sec] = w.n; if (w.overflow) J.status = -50; }
}

// token -> (bits, length)
__device__ __forceinline__ void t_tok_bits(const TexJob &J, unsigned long long tk, unsigned long long &bits, uint32_t &len) {
  const uint32_t kind = (uint32_t)(tk & 255), sym = (uint32_t)((tk >> 8) & 0xffffff), extra = (uint32_t)(tk >> 32);
  bits = 0; len = 0;
  if (kind == 255) return;
  const int mi = kind <= 1 ? 0 : (kind == 2 ? 1 : 2);
  const uint32_t s0 = kind == 1 ? 256u : (kind == 4 ? J.ns + TEX_HS : sym);
  bits = J.hm[mi].code[s0]; len = J.hm[mi].size[s0];
  if (kind == 1) { unsigned long long vb; int vl; t_vlc(extra, 4, vb, vl); bits |= vb << len; len += (uint32_t)vl; }
  else if (kind == 4) {
    bits |= (unsigned long long)J.hm[3].code[sym] << len; len += J.hm[3].size[sym];
    if (sym == 63) { unsigned long long vb; int vl; t_vlc(extra, 7, vb, vl); bits |= vb << len; len += (uint32_t)vl; }
  }
}
// grid (blocks over 3*nb slots, L): per-block bit totals
__global__ void __launch_bounds__(UVOL_BLOCK) k_pack_a(TexJob *job) {
  TexJob &J = job[blockIdx.z];
  const uint32_t l = blockIdx.y, i = blockIdx.x * UVOL_BLOCK + threadIdx.x, n = 3 * J.nb;
  unsigned long long bits; uint32_t len = 0;
  if (J.status == 0 && i < n) t_tok_bits(J, J.tok[3 * (size_t)l * J.nb + i], bits, len);
  uint32_t tot; t_block_excl_scan(len, &tot);
  if (threadIdx.x == 0) J.bsum[(size_t)l * (gridDim.x + 1) + blockIdx.x] = tot;
}
// one workgroup per slice: exclusive scan of the block totals (64-bit safe: totals < 2^32 bits per slice is asserted)
__global__ void __launch_bounds__(UVOL_BLOCK) k_pack_b(TexJob *job, uint32_t nblocks) {
  TexJob &J = job[blockIdx.z];
  const uint32_t l = blockIdx.x;
  uint32_t *bs = J.bsum + (size_t)l * (nblocks + 1);
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nblocks; b0 += UVOL_BLOCK) {
    const uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < nblocks ? bs[i] : 0, tot;
    const uint32_t ex = t_block_excl_scan(v, &tot);
    const uint32_t c = carry;
    if (i < nblocks) bs[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    bs[nblocks] = carry;
    if (J.status == 0) { J.slice_bits[l] = carry; J.slice_len[l] = (carry + 7) / 8; if ((carry + 7) / 8 + 8 > J.slice_cap) J.status = -51; }
  }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_pack_c(TexJob *job) {
  TexJob &J = job[blockIdx.z];
  const uint32_t l = blockIdx.y, i = blockIdx.x * UVOL_BLOCK + threadIdx.x, n = 3 * J.nb;
  unsigned long long bits = 0; uint32_t len = 0;
  const bool ok = J.status == 0;
  if (ok && i < n) t_tok_bits(J, J.tok[3 * (size_t)l * J.nb + i], bits, len);
  uint32_t tot;
  const uint32_t off = t_block_excl_scan(len, &tot) + (ok ? J.bsum[(size_t)l * (gridDim.x + 1) + blockIdx.x] : 0);
  if (!ok || len == 0) return;
  uint32_t *out = reinterpret_cast<uint32_t *>(J.slice[l]);
  const uint32_t wi = off >> 5, sh = off & 31;
  // up to 56 bits starting at bit `sh` of word wi: spans at most 3 words
  const unsigned long long lo = bits << sh;
  atomicOr(&out[wi], (uint32_t)lo);
  if (sh + len > 32) atomicOr(&out[wi + 1], (uint32_t)(lo >> 32));
  if (sh + len > 64) atomicOr(&out[wi + 2], (uint32_t)(bits >> (64 - sh)));
}

// ================================================================================================
// host side (K13: container)
// ================================================================================================
// K13a: payload of every segment (3 table sections + L slices, the order they have in the container) -> one packed device
// buffer, so that the batch leaves the GPU in ONE device-to-host copy (was 3 + L copies per segment: 1152 at 144 segments).
__global__ void k_tex_pack_offsets(TexJob *job, int n_seg) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long off = 0;
  for (int s = 0; s < n_seg; s++) {
    TexJob &J = job[s];
    unsigned long long len = 0;
    if (J.status == 0) { for (int k = 0; k < 3; k++) len += J.sec_len[k]; for (uint32_t l = 0; l < J.L; l++) len += J.slice_len[l]; }
    J.pack_off = off; J.pack_len = len; off += (len + 15ull) & ~15ull;
  }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_tex_pack(TexJob *job, uint8_t *packed, unsigned long long cap) {
  TJOB_OR_RETURN;
  if (J.pack_off + J.pack_len > cap) return;                           // host re-checks and reports
  uint8_t *dst = packed + J.pack_off;
  const size_t stride = (size_t)gridDim.x * UVOL_BLOCK, t0 = (size_t)blockIdx.x * UVOL_BLOCK + threadIdx.x;
  for (uint32_t p = 0; p < 3 + J.L; p++) {
    const uint8_t *src = p < 3 ? J.sec[p] : J.slice[p - 3]; const uint32_t len = p < 3 ? J.sec_len[p] : J.slice_len[p - 3];
    for (size_t i = t0; i < len; i += stride) dst[i] = src[i];
    dst += len;
  }
}

// A lane = the buffers and the stream of one batch of segments in flight.  A call on HOST inputs is cut into parts that alternate between
// the two lanes, so that the layers of part k + 1 cross PCIe (its lane's stream) while the kernels of part k run (tex_encode_segments);
// device inputs use lane 0 = the context's stream only.
struct TexLane {
  hipStream_t stream = nullptr; bool own_stream = false;
  uvol_devbuf slab, layers, job, packed;
  std::vector<TexJob> hjobs;
  uint8_t *pinned = nullptr; size_t pinned_cap = 0;
  // the part in flight (tex_submit ... tex_finish)
  int n_seg = 0, n_layers = 0, alpha = 0; uint32_t W = 0, H = 0;
  uint8_t *const *outs = nullptr; const size_t *caps = nullptr; size_t *out_lens = nullptr;
  int *status = nullptr;               // optional per-segment result codes of the part (uvol_encode_texture_segments_st)
  // the part stays in flight after tex_submit returns (busy) until tex_finish; the pointer ARRAYS of the call are copied - an enqueued call's
  // arrays are gone by the time its last part is finished on behalf of the next call (tex_flush)
  bool busy = false, on_device = false;
  std::vector<uint8_t *> outv; std::vector<size_t> capv; std::vector<const uint8_t *> srcv;
  // host layers that travel through the context's uplink (uvol_common.hpp): the slot, its generation when this part took it, and the
  // device offset of every layer in it
  UvolUpSlot *up = nullptr; uint64_t up_gen = 0; std::vector<size_t> up_off;
};
struct TexState { TexLane lane[2]; int next = 0; int deferred_rc = UVOL_OK; char deferred_err[512] = {0}; };
int tex_create(uvol_ctx *ctx) { ctx->tex = new TexState(); ctx->tex->lane[0].stream = ctx->stream; return UVOL_OK; }
void tex_destroy(uvol_ctx *ctx) {
  if (!ctx->tex) return;
  for (TexLane &t : ctx->tex->lane) {
    if (t.own_stream && t.stream) { (void)hipStreamSynchronize(t.stream); (void)hipStreamDestroy(t.stream); }
    for (uvol_devbuf *b : { &t.slab, &t.packed, &t.layers, &t.job }) if (b->p) (void)hipFree(b->p);
    if (t.pinned) (void)hipHostFree(t.pinned);
  }
  delete ctx->tex; ctx->tex = nullptr;
}
// uvol_trim: the lanes' device buffers go back to the device (nothing of the context is in flight: tex_flush has run)
int tex_trim(uvol_ctx *ctx) {
  TexState *T = ctx->tex; if (!T) return UVOL_OK;
  for (TexLane &t : T->lane) {
    if (t.busy) continue;
    if (t.stream) UVOL_HIP_CHECK(ctx, hipStreamSynchronize(t.stream));
    for (uvol_devbuf *b : { &t.slab, &t.packed, &t.layers }) if (b->p) { UVOL_HIP_CHECK(ctx, hipFree(b->p)); b->p = nullptr; b->cap = 0; }
  }
  return UVOL_OK;
}
size_t uvol_texture_bound(uint32_t w, uint32_t h, int n) {
  const size_t nb = (size_t)((w + 3) / 4) * ((h + 3) / 4);
  return 65536 + 6 * (size_t)TEX_MAX_CODEBOOK * 4 + (nb * 16 + 64) * (size_t)(n > 0 ? n : 1);      // 16 bytes per block and layer: the UASTC mode
}

namespace {
struct TCarver { size_t off = 0; template <class T> size_t take(size_t count) { off = (off + 255) & ~(size_t)255; size_t o = off; off += count * sizeof(T); return o; } };
size_t tex_layout(TexJob &J, uint8_t *base, size_t *zero_bytes) {
  TCarver C;
  const size_t NB = J.NB, nb = J.nb, KC = TEX_MAX_CODEBOOK;
#define TCARVE(field, T, count) do { size_t o_ = C.take<T>(count); if (base) field = (T *)(base + o_); } while (0)
  // ---- zeroed region ----
  TCARVE(J.hist, uint32_t, (size_t)1 << 18);
  for (int m = 0; m < TEX_NMODEL; m++) TCARVE(J.hm[m].freq, uint32_t, TEX_MODEL_CAP);
  TCARVE(J.ent, uint32_t, KC + 8); TCARVE(J.scb, uint32_t, KC + 8);
  for (uint32_t l = 0; l < J.L; l++) TCARVE(J.slice[l], uint8_t, J.slice_cap);
  *zero_bytes = (C.off + 255) & ~(size_t)255;
  // ---- rest ----
  TCARVE(J.skip, uint8_t, NB + 8); TCARVE(J.cell, uint32_t, NB + 8);
  const size_t nflag = std::max<size_t>(NB, (size_t)1 << 18) + 8;
  TCARVE(J.flag, uint8_t, nflag);
  const size_t nbs = std::max<size_t>(nflag / UVOL_BLOCK + 8, (size_t)J.L * ((3 * nb + UVOL_BLOCK - 1) / UVOL_BLOCK + 2));
  TCARVE(J.bsum, uint32_t, nbs);
  TCARVE(J.cid, uint32_t, (size_t)1 << 18); TCARVE(J.cw, uint32_t, (size_t)1 << 18); TCARVE(J.cidx, uint32_t, (size_t)1 << 18);
  for (int v = 0; v < 2; v++) {
    TexVQ &V = J.vq[v]; const size_t dim = v == 0 ? 4 : 16, ni = v == 0 ? ((size_t)1 << 18) : NB;
    TCARVE(V.leaf, uint32_t, ni + 8);
    TCARVE(V.stW, unsigned long long, KC + 8); TCARVE(V.stS, unsigned long long, (KC + 8) * dim); TCARVE(V.stQ, unsigned long long, (KC + 8) * dim);
    TCARVE(V.splittable, uint8_t, KC + 8); TCARVE(V.chosen, uint8_t, KC + 8); TCARVE(V.axis, int32_t, KC + 8);
    TCARVE(V.th, long long, KC + 8); TCARVE(V.prio, long long, KC + 8); TCARVE(V.newidx, uint32_t, KC + 8); TCARVE(V.split, uint32_t, KC + 8);
  }
  TCARVE(J.ecb, uint32_t, KC + 8); TCARVE(J.emap, uint32_t, KC + 8);
  TCARVE(J.bei, uint16_t, NB + 8); TCARVE(J.bsi, uint16_t, NB + 8); TCARVE(J.bsel, uint32_t, NB + 8);
  TCARVE(J.item, uint32_t, NB + 8);
  TCARVE(J.sused, uint8_t, KC + 8); TCARVE(J.scu, uint32_t, KC + 8); TCARVE(J.smap, uint32_t, KC + 8);
  TCARVE(J.pred, uint8_t, NB + 8);
  TCARVE(J.clist, uint32_t, NB + 8); TCARVE(J.msym, uint8_t, NB + 8); TCARVE(J.gid, uint32_t, NB + 8); TCARVE(J.gstart, uint32_t, NB + 8); TCARVE(J.gsize, uint32_t, NB + 8);
  TCARVE(J.tok, unsigned long long, 3 * NB + 8);
  for (int m = 0; m < TEX_NMODEL; m++) { TCARVE(J.hm[m].size, uint8_t, TEX_MODEL_CAP); TCARVE(J.hm[m].code, uint16_t, TEX_MODEL_CAP); }
  TCARVE(J.hscratch, uint32_t, (size_t)7 * 10 * (TEX_MODEL_CAP + 8));
  for (int s = 0; s < 3; s++) { J.sec_cap[s] = (uint32_t)(6 * KC * 4 + 4096); TCARVE(J.sec[s], uint8_t, J.sec_cap[s]); }
#undef TCARVE
  return (C.off + 255) & ~(size_t)255;
}
inline void put32(uint8_t *&p, uint32_t v) { memcpy(p, &v, 4); p += 4; }
inline void put64(uint8_t *&p, uint64_t v) { memcpy(p, &v, 8); p += 8; }
inline void put16(uint8_t *&p, uint16_t v) { memcpy(p, &v, 2); p += 2; }
}  // namespace

#define TLAUNCH(k, grid, block, shmem, ...)                                                      \
  do {                                                                                           \
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch %s\n", #k); fflush(stderr); }              \
    dim3 g3_ = grid; g3_.z = NSEG;                                                               \
    hipLaunchKernelGGL(k, g3_, block, shmem, ctx->stream, __VA_ARGS__);                          \
    if (uvol_debug()) { hipError_t e_ = hipStreamSynchronize(ctx->stream); if (e_ != hipSuccess) { fprintf(stderr, "[uvol] %s FAILED: %s\n", #k, hipGetErrorString(e_)); fflush(stderr); } } \
  } while (0)

// workgroups per segment of a statistics pass (grid-stride over the items, any count is correct): each one zeroes and flushes
// its private LDS table, so a batch of many segments takes few per segment - 512 x 144 of them spent their time on that and
// on waiting for LDS next to the geometry walkers
static inline unsigned vq_stat_blocks(unsigned item_blocks, unsigned nseg) {
  const unsigned per = std::max(32u, std::min(512u, 4096u / std::max(1u, nseg)));
  return std::min(item_blocks, per);
}
template <int DIM, int LCAP, typename CT>
static void run_vq_rounds(uvol_ctx *ctx, TexJob *dj, unsigned item_blocks, unsigned NSEG, uint32_t kmax) {
  const unsigned kb = uvol_blocks((size_t)TEX_MAX_CODEBOOK * DIM);
  const unsigned sb = vq_stat_blocks(item_blocks, NSEG);
  TLAUNCH((k_vq_zero<DIM>), dim3(kb), dim3(UVOL_BLOCK), 0, dj, 0);                // later rounds: cleared by k_vq_decide
  for (int r = 0; r < TEX_VQ_ROUNDS; r++) {
    // every leaf splits at most once per round: round r has <= 2^r leaves, so early rounds need (and reserve) little LDS and
    // no second pass - they fit next to the geometry walkers' bitmaps instead of waiting for a CU with 58 KB free
    const uint32_t leaves_r = r < 20 ? std::min<uint32_t>(kmax, 1u << r) : kmax, lds_leaves = std::min<uint32_t>((uint32_t)LCAP, std::max<uint32_t>(leaves_r, 16u));
    for (uint32_t lb = 0; lb < leaves_r; lb += LCAP) TLAUNCH((k_vq_stats<DIM, LCAP, CT>), dim3(sb), dim3(UVOL_BLOCK), (size_t)lds_leaves * (1 + 2 * DIM) * sizeof(CT), dj, 0, lb, lds_leaves);
    TLAUNCH((k_vq_decide<DIM, false>), dim3(1), dim3(UVOL_BLOCK), 0, dj, 0);
    TLAUNCH((k_vq_apply<DIM>), dim3(item_blocks), dim3(UVOL_BLOCK), 0, dj);
  }
}
template <int DIM, int LCAP, typename CT>
static void run_vq_stats(uvol_ctx *ctx, TexJob *dj, unsigned item_blocks, unsigned NSEG, uint32_t kmax) {
  const size_t shmem = (size_t)LCAP * (1 + 2 * DIM) * sizeof(CT);
  TLAUNCH((k_vq_zero<DIM>), dim3(uvol_blocks((size_t)TEX_MAX_CODEBOOK * DIM)), dim3(UVOL_BLOCK), 0, dj, 1);
  for (uint32_t lb = 0; lb < kmax; lb += LCAP) TLAUNCH((k_vq_stats<DIM, LCAP, CT>), dim3(vq_stat_blocks(item_blocks, NSEG)), dim3(UVOL_BLOCK), shmem, dj, 1, lb, (uint32_t)LCAP);
}

// selector VQ (16-D, unit weights): same rounds, statistics through k_sel_stats
static inline uint32_t sel_lcap_max() { static const uint32_t v = [] { const char *e = getenv("UVOL_SEL_LCAP"); const int x = e ? atoi(e) : 0; return (uint32_t)(x >= 16 && x <= 768 ? x : 256); }(); return v; }
static inline uint32_t sel_lcap(const TexJob &J) { const uint32_t cap = std::min<uint32_t>(sel_lcap_max(), 448u); return J.Kmax_s < cap ? J.Kmax_s : cap; }   // (448 x 17 x 8 bytes: within the 64 KiB of dynamic LDS a kernel gets without opting in)   // leaves per pass: 17 KiB of LDS, placeable next to the geometry walkers' bitmaps
// LDS words of the selector statistics are 64-bit (S | Q << 32): no bound on the items a workgroup visits, so few workgroups per
// segment and few flushes (a flush is up to 33 global atomics per leaf and workgroup).  Measured against 16-bit halves of
// 32-bit words (<= 6144 items per workgroup, 213 workgroups per 2048^2 x 5 segment): 15.9 against 27.4 ms per step.
typedef unsigned long long sel_word_t;
static inline unsigned sel_stat_blocks(const TexJob &J, unsigned nseg) { return std::min(std::max(32u, std::min(512u, 4096u / std::max(1u, nseg))), std::max(1u, (unsigned)((J.NB + UVOL_BLOCK * SEL_ILP - 1) / (UVOL_BLOCK * SEL_ILP)))); }   // (no more workgroups than trips over the blocks)
static void run_sel_stats(uvol_ctx *ctx, TexJob *dj, const TexJob &J, unsigned NSEG, int force, int round = -1) {
  // round r of the tree build has <= 2^r leaves: reserve LDS for those only (leaves past the cap would still be counted, through
  // global atomics)
  const uint32_t leaves = (round >= 0 && round < 20) ? std::min<uint32_t>(J.Kmax_s, std::max<uint32_t>(1u << round, 16u)) : J.Kmax_s;
  const uint32_t lcap = std::min<uint32_t>(sel_lcap(J), leaves);
  if (force) TLAUNCH((k_vq_zero<16>), dim3(uvol_blocks((size_t)TEX_MAX_CODEBOOK * 16)), dim3(UVOL_BLOCK), 0, dj, force);
  for (uint32_t lb = 0; lb < leaves; lb += lcap) {
    TLAUNCH(k_sel_stats<sel_word_t>, dim3(sel_stat_blocks(J, NSEG)), dim3(UVOL_BLOCK), (size_t)lcap * 17 * sizeof(sel_word_t), dj, force, lcap, lb);
  }
}
static void run_sel_rounds(uvol_ctx *ctx, TexJob *dj, const TexJob &J, unsigned item_blocks, unsigned NSEG) {
  // statistics of the root leaf (all items; force: a segment with K <= 1 is 'done' from the start and still needs its centroid),
  // then per round: decide (folds the previous round's moves into the parents) -> move the items + count the new leaves.
  // The statistics are complete after the last fold: the first Lloyd iteration reads them as they are.
  (void)item_blocks;
  run_sel_stats(ctx, dj, J, NSEG, 1, 0);
  for (int r = 0; r < TEX_VQ_ROUNDS; r++) {
    TLAUNCH((k_vq_decide<16, true>), dim3(1), dim3(UVOL_BLOCK), 0, dj, 0);
    const uint32_t m_max = r < 20 ? std::min<uint32_t>(J.Kmax_s, 1u << r) : J.Kmax_s;        // a round splits every leaf at most once
    const uint32_t lcap = std::min<uint32_t>(sel_lcap(J), std::max<uint32_t>(m_max, 16u));
    for (uint32_t wb = 0; wb < m_max; wb += lcap) {
      TLAUNCH(k_sel_split_stats<sel_word_t>, dim3(sel_stat_blocks(J, NSEG)), dim3(UVOL_BLOCK), (size_t)lcap * 17 * sizeof(sel_word_t), dj, lcap, wb);
    }
  }
  TLAUNCH((k_vq_decide<16, true>), dim3(1), dim3(UVOL_BLOCK), 0, dj, 1);
}

// n_seg segments of n_layers layers each (rgba[s * n_layers + l]), all of one size: ONE launch per stage for the whole batch
// alpha = 1: every image gets a colour and an alpha slice (basisu does this for any source image with alpha != 255; the stock
// player reads them, src/lib/KTX2Loader.js:493-497).  Called with alpha = 0 first; segments whose images turn out to have alpha
// (k_tex_skip sees every texel anyway) come back with TEX_E_ALPHA and are encoded again here with alpha = 1 - opaque
// content, the common case, pays nothing for the feature.
// first half: buffers, upload of host layers, every kernel of the batch enqueued on ctx->stream (= the lane's stream, tex_run_lane)
static int tex_submit_impl(uvol_ctx *ctx, TexLane &L, const uint8_t *const *rgba, int n_seg, int n_layers, uint32_t W, uint32_t H,
                           bool on_device, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int alpha) {
  L.n_seg = n_seg; L.n_layers = n_layers; L.alpha = alpha; L.W = W; L.H = H; L.outs = outs; L.caps = caps; L.out_lens = out_lens;
  if (n_seg <= 0) return UVOL_OK;
  if ((n_layers << alpha) > TEX_MAX_LAYERS || W > 16384 || H > 16384 || n_seg > 65535) { ctx->set_error("texture segment: unsupported size"); return UVOL_E_UNSUPPORTED; }
  const unsigned NSEG = (unsigned)n_seg;
  TexJob J0; memset(&J0, 0, sizeof(J0));
  J0.W = W; J0.H = H; J0.L = (uint32_t)n_layers << alpha; J0.ashift = (uint32_t)alpha; J0.bx = (W + 3) / 4; J0.by = (H + 3) / 4; J0.nb = J0.bx * J0.by; J0.NB = J0.nb * J0.L;
  J0.yflip = ctx->prm.y_flip ? 1 : 0;
  const int q = std::min(255, std::max(1, ctx->prm.etc1s_quality));
  J0.Kmax_e = (uint32_t)std::min(TEX_MAX_CODEBOOK, std::max(32, q * 12)); J0.Kmax_s = (uint32_t)std::min(TEX_MAX_CODEBOOK, std::max(32, q * 6));
  J0.T_skip = (uint32_t)((255 - q) * 3 / 2);
  J0.slice_cap = (uint32_t)((size_t)J0.nb * 8 + 64);
  size_t zero_bytes = 0; const size_t ws = tex_layout(J0, nullptr, &zero_bytes);
  const size_t lbytes = (size_t)W * H * 4;
  int rc;
  if ((rc = uvol_ensure(ctx, L.slab, ws * (size_t)n_seg))) return rc;
  if ((rc = uvol_ensure(ctx, L.job, sizeof(TexJob) * (size_t)n_seg))) return rc;
  const bool pre_up = !on_device && L.up != nullptr;     // the layers are already on their way (tex_encode_segments queued them on the uplink)
  if (!on_device && !pre_up && (rc = uvol_ensure(ctx, L.layers, lbytes * (size_t)n_layers * (size_t)n_seg))) return rc;
  L.hjobs.assign((size_t)n_seg, J0);
  std::vector<UvolUpItem> ups;                                           // host layers: one staged upload for the whole batch
  for (int s = 0; s < n_seg; s++) {
    TexJob &J = L.hjobs[s];
    uint8_t *base = (uint8_t *)L.slab.p + ws * (size_t)s;
    tex_layout(J, base, &zero_bytes);
    UVOL_HIP_CHECK(ctx, hipMemsetAsync(base, 0, zero_bytes, ctx->stream));
    for (int l = 0; l < n_layers; l++) {
      const uint8_t *src = rgba[(size_t)s * n_layers + l];
      if (on_device) J.layer[l] = src;
      else if (pre_up) J.layer[l] = (const uint8_t *)L.up->buf.p + L.up_off[(size_t)s * n_layers + l];
      else { uint8_t *d = (uint8_t *)L.layers.p + lbytes * ((size_t)s * n_layers + l); ups.push_back(UvolUpItem{ lbytes * ((size_t)s * n_layers + l), src, lbytes }); J.layer[l] = d; }
    }
  }
  if (!on_device && !pre_up) { const int rcu = uvol_upload_staged(ctx, (uint8_t *)L.layers.p, ups); if (rcu != UVOL_OK) return rcu; }
  if (pre_up) { const int rcu = uvol_uplink_acquire(ctx, L.up, ctx->stream); if (rcu != UVOL_OK) return rcu; }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(L.job.p, L.hjobs.data(), sizeof(TexJob) * (size_t)n_seg, hipMemcpyHostToDevice, ctx->stream));
  TexJob *dj = (TexJob *)L.job.p;
  const TexJob &J = J0;
  const unsigned bnb = uvol_blocks(J.nb), bNB = uvol_blocks(J.NB), bcell = (1u << 18) / UVOL_BLOCK, bK = uvol_blocks(TEX_MAX_CODEBOOK);
  const uint64_t src_bytes = (uint64_t)lbytes * n_layers * (uint64_t)n_seg;
  { uvol_ctx::Scope sc(ctx, "tex.k11_skip", src_bytes); TLAUNCH(k_tex_skip, dim3(bnb), dim3(UVOL_BLOCK), 0, dj);
    TLAUNCH(k_tscan_a, dim3(bNB), dim3(UVOL_BLOCK), 0, dj, J.NB);                 // coded blocks, ascending: the item list of the fit and of the selector stages
    TLAUNCH(k_tscan_b, dim3(1), dim3(UVOL_BLOCK), 0, dj, bNB);
    TLAUNCH(k_item_compact, dim3(bNB), dim3(UVOL_BLOCK), 0, dj); }
  { uvol_ctx::Scope sc(ctx, "tex.k9_endpoint_fit", src_bytes); TLAUNCH(k_tex_fit, dim3(bNB), dim3(UVOL_BLOCK), 0, dj); }
  {
    uvol_ctx::Scope sc(ctx, "tex.k10_endpoint_codebook", 0);
    TLAUNCH(k_cell_flags, dim3(bcell), dim3(UVOL_BLOCK), 0, dj);
    TLAUNCH(k_tscan_a, dim3(bcell), dim3(UVOL_BLOCK), 0, dj, 1u << 18);
    TLAUNCH(k_tscan_b, dim3(1), dim3(UVOL_BLOCK), 0, dj, bcell);
    TLAUNCH(k_cell_compact, dim3(bcell), dim3(UVOL_BLOCK), 0, dj);
    TLAUNCH(k_cell_leaf_init, dim3(bcell), dim3(UVOL_BLOCK), 0, dj);
    run_vq_rounds<4, 256, unsigned long long>(ctx, dj, bcell, NSEG, J.Kmax_e);
    for (int it = 0; it <= 2; it++) {
      run_vq_stats<4, 256, unsigned long long>(ctx, dj, bcell, NSEG, J.Kmax_e);
      TLAUNCH(k_ep_entries, dim3(bK), dim3(UVOL_BLOCK), 0, dj);
      if (it < 2) TLAUNCH(k_ep_assign, dim3(bcell), dim3(UVOL_BLOCK), 0, dj);
    }
    TLAUNCH(k_unique, dim3(1), dim3(UVOL_BLOCK), 0, dj, 0);
  }
  {
    uvol_ctx::Scope sc(ctx, "tex.k9b_block_selectors", src_bytes);
    TLAUNCH(k_block_assign, dim3(bNB), dim3(UVOL_BLOCK), 0, dj);
  }
  {
    uvol_ctx::Scope sc(ctx, "tex.k10_selector_codebook", src_bytes * 2);
    run_sel_rounds(ctx, dj, J, bNB, NSEG);
    for (int it = 0; it < 2; it++) {
      if (it) run_sel_stats(ctx, dj, J, NSEG, 1);
      TLAUNCH(k_sel_centroids, dim3(bK), dim3(UVOL_BLOCK), 0, dj);
      { uvol_ctx::Scope sc2(ctx, "tex.k10_sel_assign", 0); TLAUNCH(k_sel_assign, dim3(bNB), dim3(UVOL_BLOCK), 0, dj); }   // work (integer ops) added after the job read-back
    }
    TLAUNCH(k_sel_used, dim3(bK), dim3(UVOL_BLOCK), 0, dj, 0);
    TLAUNCH(k_sel_used, dim3(bNB), dim3(UVOL_BLOCK), 0, dj, 1);
    TLAUNCH(k_unique, dim3(1), dim3(UVOL_BLOCK), 0, dj, 1);
    TLAUNCH(k_sel_used, dim3(bNB), dim3(UVOL_BLOCK), 0, dj, 2);
    TLAUNCH(k_copy_skipped, dim3(bnb), dim3(UVOL_BLOCK), 0, dj);
  }
  const unsigned bslots = uvol_blocks((size_t)3 * J.nb);
  {
    uvol_ctx::Scope sc(ctx, "tex.k12_symbolize", (uint64_t)J.NB * 6 * n_seg);
    TLAUNCH(k_preds, dim3(bNB), dim3(UVOL_BLOCK), 0, dj);
    TLAUNCH(k_tok_delta, dim3(bNB), dim3(UVOL_BLOCK), 0, dj);
    TLAUNCH(k_sscan_a, dim3(bnb, J.L), dim3(UVOL_BLOCK), 0, dj, J.nb, J.nb);
    TLAUNCH(k_sscan_b, dim3(J.L), dim3(UVOL_BLOCK), 0, dj, bnb);
    TLAUNCH(k_coded_list, dim3(bnb, J.L), dim3(UVOL_BLOCK), 0, dj);
    const uint32_t nm = ((J.bx + 1) / 2) * ((J.by + 1) / 2); const unsigned bnm = uvol_blocks(nm);
    TLAUNCH(k_mb_flags, dim3(bnm, J.L), dim3(UVOL_BLOCK), 0, dj, nm);
    TLAUNCH(k_sscan_a, dim3(bnm, J.L), dim3(UVOL_BLOCK), 0, dj, nm, nm);
    TLAUNCH(k_sscan_b, dim3(J.L), dim3(UVOL_BLOCK), 0, dj, bnm);
    TLAUNCH(k_mb_groups, dim3(bnm, J.L), dim3(UVOL_BLOCK), 0, dj, nm);
    TLAUNCH(k_mb_tokens, dim3(bnm, J.L), dim3(UVOL_BLOCK), 0, dj, nm);
    { uvol_ctx::Scope sc2(ctx, "tex.k12_sel_tokens", (uint64_t)J.NB * 6 * n_seg); TLAUNCH(k_sel_tokens, dim3(J.L), dim3(64), 0, dj); }
    const size_t lh_bytes = (size_t)(257 + J.Kmax_e + J.Kmax_s + TEX_HS + 1 + 64) * 4;
    if (lh_bytes > 60 * 1024) { ctx->set_error("etc1s_quality too high for the LDS histogram (codebooks > 60 KiB)"); return UVOL_E_UNSUPPORTED; }
    TLAUNCH(k_tok_hist, dim3(std::min<unsigned>(uvol_blocks((size_t)3 * J.NB), 1024u)), dim3(UVOL_BLOCK), lh_bytes, dj);
  }
  {
    uvol_ctx::Scope sc(ctx, "tex.k12_huffman_pack", (uint64_t)J.NB * 24 * n_seg);
    TLAUNCH(k_huff_models, dim3(4), dim3(UVOL_BLOCK), 0, dj);
    TLAUNCH(k_sections, dim3(3), dim3(UVOL_BLOCK), 0, dj);
    TLAUNCH(k_pack_a, dim3(bslots, J.L), dim3(UVOL_BLOCK), 0, dj);
    TLAUNCH(k_pack_b, dim3(J.L), dim3(UVOL_BLOCK), 0, dj, bslots);
    TLAUNCH(k_pack_c, dim3(bslots, J.L), dim3(UVOL_BLOCK), 0, dj);
  }
  // packed payloads: sections <= caps, slices <= slice_cap each; the bound below is what the workspace itself can hold
  size_t pack_cap = 0;
  { const TexJob &Jc = L.hjobs[0]; pack_cap = (((size_t)Jc.sec_cap[0] + Jc.sec_cap[1] + Jc.sec_cap[2] + (size_t)Jc.slice_cap * (size_t)Jc.L) + 15) & ~(size_t)15; }
  // typical output is ~1 % of that bound: start from 1/8 of it and grow on demand (checked after the copy of the job records)
  size_t want_pack = std::max<size_t>(L.packed.cap, pack_cap * (size_t)n_seg / 8 + 4096);
  if (int rc = uvol_ensure(ctx, L.packed, want_pack)) return rc;
  { uvol_ctx::Scope sc(ctx, "tex.k13_pack", 0);
    hipLaunchKernelGGL(k_tex_pack_offsets, dim3(1), dim3(64), 0, ctx->stream, dj, n_seg);
    TLAUNCH(k_tex_pack, dim3(32), dim3(UVOL_BLOCK), 0, dj, (uint8_t *)L.packed.p, (unsigned long long)L.packed.cap); }
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  if (pre_up) { const int rcr = uvol_uplink_release(ctx, L.up, ctx->stream); if (rcr != UVOL_OK) return rcr; }
  return UVOL_OK;
}
// second half: waits for the lane's stream, reads the job records and the packed payloads back, writes the KTX2 containers
static int tex_finish_impl(uvol_ctx *ctx, TexLane &L) {
  const int n_seg = L.n_seg, n_layers = L.n_layers, alpha = L.alpha; const uint32_t W = L.W, H = L.H;
  uint8_t *const *outs = L.outs; const size_t *caps = L.caps; size_t *out_lens = L.out_lens;
  if (n_seg <= 0) return UVOL_OK;
  TexJob *dj = (TexJob *)L.job.p; const unsigned NSEG = (unsigned)n_seg;
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  UVOL_HIP_CHECK(ctx, hipMemcpy(L.hjobs.data(), dj, sizeof(TexJob) * (size_t)n_seg, hipMemcpyDeviceToHost));
  if (ctx->profiling) {            // matrix-core work of the two k_sel_assign launches: items x entries (padded to tiles of 16) x 64-long dot x 2 planes x 2 ops
    uint64_t ops = 0;
    for (int s = 0; s < n_seg; s++) { const TexVQ &V = L.hjobs[s].vq[1]; ops += 2ull * (uint64_t)V.n_items * (uint64_t)((V.nl + 15u) & ~15u) * 64ull * 2ull * 2ull; }
    ctx->prof[ctx->prof_index("tex.k10_sel_assign")].algo_bytes += ops;
  }
  size_t packed_total = 0;
  for (int s = 0; s < n_seg; s++) packed_total = std::max<size_t>(packed_total, (size_t)(L.hjobs[s].pack_off + ((L.hjobs[s].pack_len + 15ull) & ~15ull)));
  if (packed_total > L.packed.cap) {                                  // rare: grow and gather again
    if (int rc = uvol_ensure(ctx, L.packed, packed_total)) return rc;
    TLAUNCH(k_tex_pack, dim3(32), dim3(UVOL_BLOCK), 0, dj, (uint8_t *)L.packed.p, (unsigned long long)L.packed.cap);
    UVOL_HIP_CHECK(ctx, hipGetLastError());
  }
  if (packed_total > L.pinned_cap) {
    if (L.pinned) (void)hipHostFree(L.pinned);
    L.pinned = nullptr; L.pinned_cap = 0;
    const size_t want = packed_total + packed_total / 4 + 4096;
    UVOL_HIP_CHECK(ctx, hipHostMalloc((void **)&L.pinned, want, hipHostMallocDefault));
    L.pinned_cap = want;
  }
  if (packed_total) UVOL_HIP_CHECK(ctx, hipMemcpyAsync(L.pinned, L.packed.p, packed_total, hipMemcpyDeviceToHost, ctx->stream));
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  // ---- K13: KTX2 containers ----
  static const uint8_t ident[12] = { 0xAB, 'K', 'T', 'X', ' ', '2', '0', 0xBB, '\r', '\n', 0x1A, '\n' };
  static const char writer[] = "uvol-mi355x etc1s 0.1";
  uint8_t kvd[128]; uint8_t *kp = kvd;
  put32(kp, 12 + 12); memcpy(kp, "KTXanimData", 12); kp += 12; put32(kp, 1); put32(kp, 15); put32(kp, 0);
  put32(kp, 10 + (uint32_t)sizeof(writer)); memcpy(kp, "KTXwriter", 10); kp += 10; memcpy(kp, writer, sizeof(writer)); kp += sizeof(writer);
  while ((kp - kvd) & 3) *kp++ = 0;
  const uint32_t dfd_off = 80 + 24, dfd_len = alpha ? 60 : 44, kvd_off = dfd_off + dfd_len, kvd_len = (uint32_t)(kp - kvd);
  const int nsl = n_layers << alpha;
  const uint64_t sgd_off = ((uint64_t)kvd_off + kvd_len + 7) & ~7ull;
  int worst = UVOL_OK;
  for (int s = 0; s < n_seg; s++) {
    const TexJob &R = L.hjobs[s];
    if (L.status) L.status[s] = UVOL_OK;
    if (R.status == TEX_E_ALPHA && !alpha) { out_lens[s] = TEX_RETRY_ALPHA; continue; }                    // encoded again with alpha slices by the caller
    if (R.status != 0) { ctx->set_error("texture segment %d: device status %d", s, R.status); worst = UVOL_E_ENCODE; out_lens[s] = 0; if (L.status) L.status[s] = UVOL_E_ENCODE; continue; }
    const uint64_t sgd_len = 20 + 20 * (uint64_t)n_layers + R.sec_len[0] + R.sec_len[1] + R.sec_len[2];
    uint64_t lvl_len = 0; for (int l = 0; l < nsl; l++) lvl_len += R.slice_len[l];
    const uint64_t lvl_off = sgd_off + sgd_len, total = lvl_off + lvl_len;
    out_lens[s] = (size_t)total;
    if (total > caps[s]) { ctx->set_error("texture segment %d: output buffer too small (%llu > %llu)", s, (unsigned long long)total, (unsigned long long)caps[s]); worst = UVOL_E_NOSPACE; if (L.status) L.status[s] = UVOL_E_NOSPACE; continue; }
    uint8_t *out = outs[s], *p = out;
    memcpy(p, ident, 12); p += 12;
    put32(p, 0); put32(p, 1); put32(p, W); put32(p, H); put32(p, 0); put32(p, (uint32_t)n_layers); put32(p, 1); put32(p, 1); put32(p, 1);
    put32(p, dfd_off); put32(p, dfd_len); put32(p, kvd_off); put32(p, kvd_len); put64(p, sgd_off); put64(p, sgd_len);
    put64(p, lvl_off); put64(p, lvl_len); put64(p, 0);
    put32(p, dfd_len); put32(p, 0); put16(p, 2); put16(p, (uint16_t)(dfd_len - 4));
    *p++ = 163; *p++ = 1; *p++ = 2; *p++ = 0; *p++ = 3; *p++ = 3; *p++ = 0; *p++ = 0;
    for (int i = 0; i < 8; i++) *p++ = 0;
    put16(p, 0); *p++ = 63; *p++ = 0; *p++ = 0; *p++ = 0; *p++ = 0; *p++ = 0; put32(p, 0); put32(p, 0xFFFFFFFFu);
    if (alpha) { put16(p, 64); *p++ = 63; *p++ = 15; *p++ = 0; *p++ = 0; *p++ = 0; *p++ = 0; put32(p, 0); put32(p, 0xFFFFFFFFu); }      // second sample: channel 15 (AAA) at bit 64
    memcpy(p, kvd, kvd_len); p += kvd_len;
    while ((uint64_t)(p - out) < sgd_off) *p++ = 0;
    put16(p, (uint16_t)R.ne); put16(p, (uint16_t)R.ns); put32(p, R.sec_len[0]); put32(p, R.sec_len[1]); put32(p, R.sec_len[2]); put32(p, 0);
    { uint32_t off = 0;
      for (int l = 0; l < n_layers; l++) {
        const uint32_t c = R.slice_len[l << alpha], a = alpha ? R.slice_len[(l << alpha) + 1] : 0u;
        put32(p, l > 0 ? 2 : 0); put32(p, off); put32(p, c); put32(p, alpha ? off + c : 0u); put32(p, a); off += c + a; } }
    if (R.pack_len != sgd_len - 20 - 20 * (uint64_t)n_layers + lvl_len) { ctx->set_error("texture segment %d: packed payload length mismatch", s); worst = UVOL_E_ENCODE; if (L.status) L.status[s] = UVOL_E_ENCODE; continue; }
    memcpy(p, L.pinned + R.pack_off, (size_t)R.pack_len);            // sections then slices, already in container order
  }
  return L.status ? UVOL_OK : worst;      // (with status[] the per-segment codes carry the failures)
}
// the lane's stream stands in for the context's while one of its halves runs (TLAUNCH, Scope, uvol_ensure, uvol_upload_staged use ctx->stream)
static int tex_submit(uvol_ctx *ctx, TexLane &L, const uint8_t *const *rgba, int n_seg, int n_layers, uint32_t W, uint32_t H,
                      bool on_device, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int alpha, int *status = nullptr) {
  hipStream_t saved = ctx->stream; ctx->stream = L.stream; L.status = status; L.on_device = on_device;
  L.outv.assign(outs, outs + n_seg); L.capv.assign(caps, caps + n_seg); L.srcv.assign(rgba, rgba + (size_t)n_seg * n_layers);
  const int rc = tex_submit_impl(ctx, L, L.srcv.data(), n_seg, n_layers, W, H, on_device, L.outv.data(), L.capv.data(), out_lens, alpha);
  if (rc != UVOL_OK && L.up) { (void)hipStreamSynchronize(L.stream); L.up = nullptr; }      // (kernels already enqueued may read the slot: its release was never recorded)
  ctx->stream = saved; L.busy = rc == UVOL_OK; return rc;
}
// finish + the second pass of the segments that turned out to have alpha (one more batch on the same lane; host layers were uploaded
// by the first pass and are read where they lie - unless they travelled through an uplink slot that has been filled again since)
static int tex_finish(uvol_ctx *ctx, TexLane &L) {
  if (!L.busy) return UVOL_OK;
  L.busy = false;
  hipStream_t saved = ctx->stream; ctx->stream = L.stream;
  const bool on_device = L.on_device; const uint8_t *const *rgba = L.srcv.data();
  const int n_seg = L.n_seg, n_layers = L.n_layers; uint8_t *const *outs = L.outs; const size_t *caps = L.caps; size_t *out_lens = L.out_lens; const uint32_t W = L.W, H = L.H;
  int *const status = L.status;
  int rc = tex_finish_impl(ctx, L);
  std::vector<int> again;
  for (int s = 0; s < n_seg; s++) if (out_lens[s] == TEX_RETRY_ALPHA) { again.push_back(s); out_lens[s] = 0; }
  if (!again.empty() && (rc == UVOL_OK || rc == UVOL_E_NOSPACE || rc == UVOL_E_ENCODE)) {
    UvolUpSlot *const slot = L.up; const bool slot_live = slot && slot->gen == L.up_gen;      // (one thread drives the context: nothing fills the slot during this function)
    const bool from_host = !on_device && slot && !slot_live;                                  // its bytes are gone: the layers cross the link once more, from the caller's arrays
    std::vector<const uint8_t *> src; std::vector<uint8_t *> o2; std::vector<size_t> c2, l2(again.size(), 0); std::vector<int> st2(again.size(), UVOL_OK);
    for (int s : again) { for (int l = 0; l < n_layers; l++) src.push_back((on_device || from_host) ? rgba[(size_t)s * n_layers + l] : L.hjobs[s].layer[l]); o2.push_back(outs[s]); c2.push_back(caps[s]); }
    L.status = status ? st2.data() : nullptr;
    L.up = nullptr;                                        // (the re-run reads device pointers, or uploads through the lane's own buffer)
    int rc2 = tex_submit_impl(ctx, L, src.data(), (int)again.size(), n_layers, W, H, !from_host, o2.data(), c2.data(), l2.data(), 1);
    if (rc2 == UVOL_OK && slot_live) rc2 = uvol_uplink_release(ctx, slot, ctx->stream);      // the slot is read until here
    if (rc2 == UVOL_OK) rc2 = tex_finish_impl(ctx, L); else (void)hipStreamSynchronize(ctx->stream);
    L.status = status;
    for (size_t i = 0; i < again.size(); i++) { out_lens[again[i]] = l2[i]; if (status) status[again[i]] = rc2 != UVOL_OK ? rc2 : st2[i]; }
    if (rc == UVOL_OK) rc = rc2;
  }
  L.up = nullptr;
  ctx->stream = saved; return rc;
}
// completes the parts still in flight on the lanes (an enqueued call leaves its last part for the next call, or for this), older part first;
// returns the first error among them and among parts finished earlier on behalf of later calls
int tex_flush(uvol_ctx *ctx) {
  TexState *T = ctx->tex; if (!T) return UVOL_OK;
  int rc = T->deferred_rc; T->deferred_rc = UVOL_OK;
  if (rc != UVOL_OK) snprintf(ctx->err, sizeof ctx->err, "%s", T->deferred_err);
  for (int k = 0; k < 2; k++) { const int r = tex_finish(ctx, T->lane[(T->next + k) & 1]); if (rc == UVOL_OK) rc = r; }
  ctx->resolve_profile();
  return rc;
}
// segments per part of a call on host inputs (UVOL_TEX_PART, tests: small values cut small calls too; 0 = never cut)
static inline int tex_part_segments() { static const int v = [] { const char *e = getenv("UVOL_TEX_PART"); const int k = e ? atoi(e) : 64; return k < 0 ? 0 : k; }(); return v; }
// defer = the enqueue form: the call's LAST part stays in flight when the call returns and is finished when the next call needs its lane
// (or by tex_flush when the worker's queue runs empty), so that consecutive enqueued calls overlap on the device - the next call's first
// part uploads and encodes beside this call's last
int tex_encode_segments(uvol_ctx *ctx, const uint8_t *const *rgba, int n_seg, int n_layers, uint32_t W, uint32_t H,
                        bool on_device, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status, bool defer) {
  if (n_seg <= 0) return UVOL_OK;
  TexState *T = ctx->tex;
  T->lane[0].stream = ctx->stream;
  if (on_device) { const int ro = png_order_before(ctx, ctx->stream, rgba, (size_t)n_seg * n_layers); if (ro != UVOL_OK) return ro; }      // layers un-filtered on the ingest stream (uvol_unfilter_png_batch_dev)
  // device inputs: parts of UVOL_TEX_PART_DEV segments (default 128) - a segment in flight holds ~158 MB of workspace (2048^2 x 5 layers), so
  // the 512 segments of a 2560-frame pass held 81 GB at once; in parts on the two lanes 40 GB, which the geometry context's frames in
  // flight are worth more than (profiles/r05_frames_in_flight.json)
  static const int part_dev = [] { const char *e = getenv("UVOL_TEX_PART_DEV"); const int k = e ? atoi(e) : 128; return k < 0 ? 0 : k; }();
  const int part = on_device ? part_dev : tex_part_segments();
  const bool single = part <= 0 || n_seg < 2 * part;
  const int parts = single ? 1 : (on_device ? (n_seg + part - 1) / part : std::max(2, std::min(4, n_seg / part)));           // few, large parts: every part pays the serial stages' latency (one wave per slice) once
  auto lo = [&](int k) { return (int)((long long)n_seg * k / parts); };
  // Layers in uvol_host_alloc memory: the uploads of ALL parts are queued on the context's copy stream now, one uplink slot per part
  // (uvol_common.hpp "Uplink"; at least four slots, so a part's slot was last read by a part whose kernels have been enqueued)
  struct PartUp { UvolUpSlot *slot = nullptr; std::vector<size_t> off; };
  std::vector<PartUp> pups;
  if (!on_device && uvol_uplink_enabled()) {
    const size_t lbytes = (size_t)W * H * 4; bool all = true;
    for (size_t i = 0; i < (size_t)n_seg * n_layers && all; i++) all = rgba[i] && uvol_host_pinned(rgba[i], lbytes);
    UvolUplink *U = all ? uvol_uplink(ctx, (size_t)std::max(2 + uvol_uplink_ahead(), parts)) : nullptr;
    if (all && !U) return UVOL_E_HIP;
    if (U) {
      pups.resize((size_t)parts);
      std::vector<UvolUpItem> items;
      for (int k = 0; k < parts; k++) {
        const int a = lo(k), b = lo(k + 1);
        PartUp &P = pups[(size_t)k]; P.off.resize((size_t)(b - a) * n_layers);
        UvolUpPlacer pl; items.clear(); items.reserve(P.off.size());
        for (size_t i = 0; i < P.off.size(); i++) { const uint8_t *src = rgba[(size_t)a * n_layers + i]; const size_t d = pl.place(src, lbytes); P.off[i] = d; items.push_back(UvolUpItem{ d, src, lbytes }); }
        P.slot = uvol_uplink_fill(ctx, U, items, pl.total());
        if (!P.slot) { (void)hipStreamSynchronize(U->stream); return UVOL_E_HIP; }
      }
    }
  }
  auto take_up = [&](TexLane &L, int k) { if (pups.empty()) { L.up = nullptr; return; } L.up = pups[(size_t)k].slot; L.up_gen = L.up->gen; L.up_off = std::move(pups[(size_t)k].off); };
  auto fail_ups = [&]() { if (!pups.empty() && ctx->uplink) (void)hipStreamSynchronize(ctx->uplink->stream); };      // the copies nobody will read must not outlive the caller's arrays
  auto keep_err = [&](int r) { if (r != UVOL_OK && T->deferred_rc == UVOL_OK) { T->deferred_rc = r; snprintf(T->deferred_err, sizeof T->deferred_err, "%s", ctx->err); } };
  int rc = UVOL_OK;
  if (single) {                                                           // one batch on the context's stream
    keep_err(tex_flush(ctx));                                             // (nothing of an earlier enqueued call on the lanes)
    take_up(T->lane[0], 0);
    rc = tex_submit(ctx, T->lane[0], rgba, n_seg, n_layers, W, H, on_device, outs, caps, out_lens, 0, status);
    if (rc != UVOL_OK) { fail_ups(); return rc; }
    T->next = 1;                                                          // (lane 0 holds the older part)
    if (defer) return UVOL_OK;
    return tex_flush(ctx);
  }
  // A large call: parts of >= `part` segments alternate between two lanes.  While the GPU encodes part k the layers of part k + 1 cross
  // the link (staged through the pinned buffers by this thread, or queued on the uplink above) and its kernels are enqueued on the other
  // lane's stream; then part k's containers are written while part k + 1 encodes.  16.8 MB per layer cross PCIe: without the overlap a
  // call was upload, then encode, one after the other.
  if (!T->lane[1].stream) { if (uvol_make_stream(ctx, &T->lane[1].stream) != hipSuccess) { ctx->set_error("texture lane: stream creation failed"); fail_ups(); return UVOL_E_HIP; } T->lane[1].own_stream = true; }
  if (on_device) { const int ro = png_order_before(ctx, T->lane[1].stream, rgba, (size_t)n_seg * n_layers); if (ro != UVOL_OK) return ro; }      // (the second lane reads the un-filtered layers too)
  for (int k = 0; k < parts; k++) {
    TexLane &L = T->lane[T->next & 1], &O = T->lane[(T->next & 1) ^ 1];
    keep_err(tex_finish(ctx, L));                                         // (two parts in flight at most: normally a no-op, the loop below finished it)
    const int a = lo(k), b = lo(k + 1);
    take_up(L, k);
    const int r = tex_submit(ctx, L, rgba + (size_t)a * n_layers, b - a, n_layers, W, H, on_device, outs + a, caps + a, out_lens + a, 0, status ? status + a : nullptr);
    if (r != UVOL_OK) { fail_ups(); (void)tex_flush(ctx); return r; }
    // the part before this one (of this call, or the last part of the call before) - but an enqueued call returns with its last TWO parts in flight:
    // the next call's uploads are queued when it begins, and a worker that waited here for part parts - 2 reached that point after the link had run dry
    if (!(defer && k == parts - 1)) keep_err(tex_finish(ctx, O));
    T->next ^= 1;
  }
  if (defer) return UVOL_OK;
  return tex_flush(ctx);
}

int tex_encode_segment(uvol_ctx *ctx, const uint8_t *const *rgba, int n_layers, uint32_t W, uint32_t H,
                       bool on_device, uint8_t *out, size_t cap, size_t *out_len) {
  return tex_encode_segments(ctx, rgba, 1, n_layers, W, H, on_device, &out, &cap, out_len, nullptr);
}
