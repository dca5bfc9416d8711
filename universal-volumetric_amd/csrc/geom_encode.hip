// geom_encode.hip — hand-written HIP (gfx950) geometry encoder: OBJ-shaped arrays -> Draco 2.2 .drc.
//
// Replaces the arithmetic of HOT LOOP 1 of the reference (scripts/Encoder.py:256-267, one
// `draco_encoder -qp 11 -qt 10 -qn 8 -cl 7` process per frame) for a whole batch of frames at once.
// Kernel groups: K1 min/max+quantise, K2 value dedup, K3 corner table,
// K4 valence edgebreaker, K5 attribute DFS order, K6 prediction residuals, K7 rANS/rabs.
// Every kernel takes the device array of GeoJob and uses blockIdx.y (or .z) as the frame index, so
// a batch is ONE launch per stage: the parallel stages fill the chip, the serial walkers run one
// frame (or one entropy stream) per workgroup concurrently.
//
// There is no MFMA here by design: the path is integer/byte work bounded by dependent-load latency
// (walkers) and HBM/L2 bandwidth (parallel stages).
#include <chrono>
#include <thread>
#include "uvol_common.hpp"
#include "geom_device.hpp"
#include "uvol_ws.hpp"
#include <algorithm>
#include <map>

#define JOB_OR_RETURN GeoJob &J = jobs[blockIdx.y]; if (J.status != 0) return
// For kernels with barriers: the frame's status is read by ONE thread and the whole workgroup takes the same decision.  Another
// workgroup of the same frame may fail the frame at any moment; with a per-thread test some waves of a block would leave and
// the others wait for them at the barrier (the hardware tolerates that, the host emulation of tests/hipemu does not).
#define JOB_OR_RETURN_UNIFORM GeoJob &J = jobs[blockIdx.y]; { __shared__ int job_st_; if (threadIdx.x == 0) job_st_ = J.status; __syncthreads(); if (job_st_ != 0) return; }

#include "geo_scan.hpp"
#include "geo_dedup.hpp"
#include "geo_faces.hpp"
#include "geo_corner_table.hpp"
#include "geo_walk_lds.hpp"
#include "geo_conn.hpp"
#include "geo_traverse_lds.hpp"
#include "geo_walk_simt.hpp"
#include "geo_attr.hpp"
#include "geo_seq.hpp"
#include "geo_entropy.hpp"
#include "geo_layout.hpp"
// ================================================================================================
// host side
// ================================================================================================
// workspace placement (see ws_collect / ws_place below)
struct WsItem { size_t slot, bytes; int first, last; size_t off; };
struct WsPlan {
  std::vector<uint64_t> key; std::vector<size_t> offs; size_t total = 0, zero = 0;
};
// placements by (bucketed) frame shape: the frames of a capture all differ a little in their counts (the reference's 250 frames have
// 26,144 - 27,979 vertices), and a first-fit placement per frame (~100 us, twice) would cost the host more than the GPU needs to encode
struct WsPlanCache { std::map<std::vector<uint64_t>, WsPlan> plans; };
// A lane = everything ONE group of frames needs while it is in flight: two streams (main + auxiliary), its events, its device
// buffers and the host-side record of the call it belongs to.  A call is cut into groups that run on different lanes, so that the
// bandwidth-bound front end of one group runs beside the latency-bound walkers of another (geo_encode_batch); lane 0 runs on the
// context's own stream.
// a group's inputs already on their way through the context's uplink (uvol_common.hpp): the slot and the device offset of each of the
// frame's six arrays in it (pos, uv, nrm, idx_pos, idx_uv, idx_nrm; unused ones are never read)
struct GeoUp { UvolUpSlot *slot = nullptr; std::vector<size_t> off; };
struct GeoLane {
  GeoUp up;               // the group in flight reads its inputs from this uplink slot (slot == nullptr: caller's device arrays, or `inputs` below)
  hipStream_t stream = nullptr, aux = nullptr; bool own_stream = false;     // aux: valence replay runs beside renumber / seams / traversals
  hipEvent_t ev_walk = nullptr, ev_val = nullptr, ev_fe = nullptr;          // ev_fe: this group's front end (dedup + corner table) is done
  uvol_devbuf slab;       // all per-job workspaces
  uvol_devbuf inputs;     // staged inputs when the caller passes host pointers
  uvol_devbuf jobs;       // GeoJob[n]
  uvol_devbuf outs;       // output buffers
  std::vector<GeoJob> hjobs;
  uint8_t *pinned = nullptr; size_t pinned_cap = 0;
  uint32_t *counts = nullptr;          // device: {frames relabelled, frames with their predecessor's connectivity} of the group (k_relabel_decide)
  // the group in flight (submitted, not completed): its slice of the caller's arrays (the pointer arrays are copied: an enqueued call's arrays are gone by then)
  bool busy = false, on_device = false, full = false;
  std::vector<uvol_mesh> meshes; std::vector<uint8_t *> outp; std::vector<size_t> caps;
  size_t *out_lens = nullptr; int *status = nullptr; int n = 0, n_conc = 0;
  // GPU-resident form (uvol_encode_mesh_batch_dev_out): the packed output area is the CALLER's device buffer and the payload is not copied out
  uint8_t *ext_out = nullptr; size_t ext_cap = 0; size_t *ext_offs = nullptr; hipStream_t producer = nullptr; hipEvent_t ev_prod = nullptr;
  std::chrono::steady_clock::time_point t_enter; double t_prep = 0, t_enq = 0;
};
struct GeoState {
  std::vector<GeoLane *> lanes; int next_lane = 0;
  int lanes_cap = 1 << 20;             // lanes the ring may use: lowered when a lane could not get its workspace (three lanes of general-layout groups do not fit beside a full set of inputs), reset by uvol_trim
  hipEvent_t walk_last = nullptr;      // walk event of the group submitted last (UVOL_GEO_CHAIN=2)
  hipEvent_t fe_last = nullptr;        // front-end event of the group submitted last (the front ends of consecutive groups run one after the other)
  int deferred_rc = UVOL_OK;           // first error among groups completed on behalf of a later call (geo_flush returns it)
  WsPlanCache plan; std::vector<WsItem> items;     // workspace placements by frame shape
  size_t max_lds = 64 * 1024;
  int num_cu = 256;                    // CUs this context's streams may run on
  bool blocking_call = false;          // the group being submitted belongs to a BLOCKING call (geo_encode_batch): nothing else of this context runs beside it
  bool compact_ok = true;              // the last group was all clean, coherently stored frames: the next one starts in the compact layout (geo_submit_impl)
};
static void geo_lane_free(GeoLane *L) {
  if (L->aux) { (void)hipStreamSynchronize(L->aux); (void)hipStreamDestroy(L->aux); }
  if (L->own_stream && L->stream) { (void)hipStreamSynchronize(L->stream); (void)hipStreamDestroy(L->stream); }
  for (uvol_devbuf *b : { &L->slab, &L->inputs, &L->jobs, &L->outs }) if (b->p) (void)hipFree(b->p);
  if (L->pinned) (void)hipHostFree(L->pinned);
  for (hipEvent_t e : { L->ev_walk, L->ev_val, L->ev_fe, L->ev_prod }) if (e) (void)hipEventDestroy(e);
  if (L->counts) (void)hipFree(L->counts);
  delete L;
}
// lane k of the context (created on first use; lane 0 = the context's stream)
static GeoLane *geo_lane(uvol_ctx *ctx, int k) {
  GeoState *G = ctx->geo;
  while ((int)G->lanes.size() <= k) {
    GeoLane *L = new GeoLane();
    if (G->lanes.empty()) L->stream = ctx->stream; else { if (uvol_make_stream(ctx, &L->stream) != hipSuccess) { delete L; return nullptr; } L->own_stream = true; }
    if (uvol_make_stream(ctx, &L->aux) != hipSuccess || hipEventCreate(&L->ev_walk) != hipSuccess || hipEventCreate(&L->ev_val) != hipSuccess ||
        hipEventCreateWithFlags(&L->ev_fe, hipEventDisableTiming) != hipSuccess) { geo_lane_free(L); return nullptr; }
    G->lanes.push_back(L);
  }
  return G->lanes[k];
}

// uvol_trim: the device workspaces of every lane go back to the device (they only grow: a lane that once ran a whole 2560-frame call keeps
// 130 GB); streams, events and the small job arrays stay.  The caller has completed the context's work (geo_flush).
int geo_trim(uvol_ctx *ctx) {
  GeoState *G = ctx->geo; if (!G) return UVOL_OK;
  G->lanes_cap = 1 << 20;
  for (GeoLane *L : G->lanes) {
    if (L->busy) continue;
    UVOL_HIP_CHECK(ctx, hipStreamSynchronize(L->stream)); UVOL_HIP_CHECK(ctx, hipStreamSynchronize(L->aux));
    for (uvol_devbuf *b : { &L->slab, &L->inputs, &L->outs }) if (b->p) { UVOL_HIP_CHECK(ctx, hipFree(b->p)); b->p = nullptr; b->cap = 0; }
  }
  return UVOL_OK;
}
int geo_create(uvol_ctx *ctx) {
  ctx->geo = new GeoState();
  if (!geo_lane(ctx, 0)) return UVOL_E_HIP;
#ifndef HIPEMU
  // the serial walkers keep their visited bitmaps in LDS: allow the full 160 KiB of a gfx950 CU
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) == hipSuccess && v > 0) ctx->geo->max_lds = (size_t)v;
  const size_t want = ctx->geo->max_lds;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && v > 0) ctx->geo->num_cu = v;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eb_walk<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eb_walk<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_traverse<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
  (void)hipGetLastError();
#else
  ctx->geo->max_lds = 160 * 1024;
#endif
  return UVOL_OK;
}
void geo_destroy(uvol_ctx *ctx) {
  if (!ctx->geo) return;
  GeoState *g = ctx->geo;
  for (GeoLane *L : g->lanes) geo_lane_free(L);
  delete g; ctx->geo = nullptr;
}

// entry capacity of a frame's per-vertex arrays = size of its vertex id space (ws_collect sizes them with it; geo_rec8 bounds the record fields with it)
static inline uint64_t geo_ecap(uint32_t n_pos, uint32_t n_uv, uint32_t n_nrm, uint32_t nf, bool full) {
  const uint64_t vmax = std::max<uint64_t>(n_pos, std::max<uint64_t>(n_uv, n_nrm)), nc = 3ull * nf;
  return full ? vmax + 2 * nc + 3 : std::min(vmax + 2 * nc + 3, vmax + vmax / 2 + 4096);
}
namespace {
inline uint32_t pow2_at_least(uint64_t v) { uint32_t c = 16; while (c < v) c <<= 1; return c; }

// ------------------------------------------------------------------------------------------------
// Workspace of one frame.  Every array has a lifetime [first, last] in pipeline phases; arrays whose lifetimes do not
// overlap share addresses (greedy first-fit over the live intervals, largest first), so a 200 k-face frame needs ~55 MB
// instead of 220 MB and a 288 GB GPU holds > 2000 frames in flight.  Arrays that must start out zero are pinned at the head
// of the workspace (one k_job_clear launch per batch) and are never shared.
// Per-vertex / per-entry arrays are sized for `ecap` entries (1.5 x the largest input attribute + slack) instead of the
// worst case 3 * faces; a mesh that needs more (non-manifold fans, every corner its own vertex) fails with GEO_E_WS_OVERFLOW
// on the device and is re-encoded alone with worst-case sizes (geo_encode_batch), so the compact layout never costs correctness.
// Phases (main stream order; the auxiliary stream runs events / valence replay / context scatter between PH_FTIME and PH_HIST):
enum { PH_DEDUP = 0, PH_FACES, PH_CT, PH_FANS0, PH_DENSE0, PH_WALK, PH_FTIME, PH_RENUM, PH_SEAMS, PH_DENSE1, PH_TRAV, PH_V2D, PH_QUANT,
       PH_PRED, PH_HIST, PH_ENT, PH_LAYOUT, PH_PINNED = -1 };

// Collects the arrays of job J (sizes from its input counts) and sets the capacities stored in J.  full = worst-case sizes.
// fmt0 / fmtT: record format of the walk table / of the three traversal tables (pack_face_records: 0, 1, 2)
void ws_collect(GeoJob &J, bool full, int fmt0, int fmtT, std::vector<WsItem> &items) {
  items.clear();
  const size_t nfi = J.nf_in, nc = 3 * nfi;
  const size_t vmax = std::max<size_t>(J.n_pos, std::max<size_t>(J.n_uv, J.n_nrm));
  // vertex ids are position ids + one id per further fan of a non-manifold position + (attribute tables) one per seam segment
  const size_t ecap = (size_t)geo_ecap(J.n_pos, J.n_uv, J.n_nrm, J.nf_in, full);
  J.ecap = (uint32_t)ecap;
  auto bitlen = [](uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; };
#define CARVE(field, T, count, first, last) items.push_back(WsItem{(size_t)((char *)&(field) - (char *)&J), (size_t)(count) * sizeof(T), (first), (last), 0})
  // dedup scratch, live in the first phase only.  Compact layout: the partitioned form (records by hash bin + the counts
  // matrix); worst-case layout (retries): the hash tables, zeroed right before use (k_dd_clear)
  for (int k = 0; k < 3; k++) {
    const uint32_t n = k == 0 ? J.n_pos : (k == 1 ? J.n_uv : J.n_nrm);
    J.dd_cap[k] = pow2_at_least(2ull * n + 2);
    J.dd_nblk[k] = (uint32_t)((n + DD_TILE - 1) / DD_TILE);
    J.dd_nb[k] = (uint32_t)std::min<uint64_t>(DD_MAXBINS, pow2_at_least(std::max<uint64_t>(1, n / 1024)));
    if (full) { CARVE(J.dd_tab[k], uint32_t, J.dd_cap[k], PH_DEDUP, PH_DEDUP); }
    else { CARVE(J.dd_part[k], uint4, (size_t)n + 1, PH_DEDUP, PH_DEDUP); CARVE(J.dd_cnt[k], uint32_t, (size_t)J.dd_nb[k] * J.dd_nblk[k] + 2, PH_DEDUP, PH_DEDUP); }
  }
  // locality relabelling (k_ms_*): keys, {key, index} records, counts matrices; new position ids / positions in that order; face maps
  if (J.relabel && !J.compact) {
    const size_t nmax = std::max<size_t>(J.n_pos, nfi);
    auto bits_of = [&](uint64_t v) { uint32_t b = 0; while (v) { b++; v >>= 1; } return b; };
    // bins of the first level: ~512 keys each, at most MS_MAXBINS (30-bit Morton keys / ids below n_pos: bin = the key's top bits)
    const uint32_t lb0 = std::min<uint32_t>(10, bits_of(J.n_pos / 512)), lb1 = std::min<uint32_t>(10, bits_of(nfi / 512));
    J.ms_sh[0] = 30 - lb0; J.ms_nb[0] = 1u << lb0; J.ms_nblk[0] = (uint32_t)((J.n_pos + MS_TILE - 1) / MS_TILE);
    { const uint32_t kb = bits_of(J.n_pos ? J.n_pos - 1 : 0); J.ms_sh[1] = kb > lb1 ? kb - lb1 : 0; }
    J.ms_nb[1] = 0; J.ms_nblk[1] = 0;                      // set by k_relabel_decide
    const uint32_t ms_nblk1 = (uint32_t)((nfi + MS_TILE - 1) / MS_TILE);
    CARVE(J.ms_key[0], uint32_t, (size_t)J.n_pos + 1, PH_DEDUP, PH_DEDUP); CARVE(J.ms_key[1], uint32_t, nfi + 1, PH_FACES, PH_FACES);
    CARVE(J.ms_part, uint2, nmax + 1, PH_DEDUP, PH_FACES);
    CARVE(J.ms_cnt, uint32_t, (size_t)MS_MAXBINS * std::max(J.ms_nblk[0], ms_nblk1) + 2, PH_DEDUP, PH_FACES);
    CARVE(J.prank, uint32_t, (size_t)J.n_pos + 1, PH_DEDUP, PH_FACES); CARVE(J.pos_s, float, 3 * (size_t)J.n_pos + 3, PH_DEDUP, PH_QUANT);
    CARVE(J.fperm, uint32_t, nfi + 1, PH_FACES, PH_FACES); CARVE(J.cidx, uint32_t, nfi + 1, PH_FACES, PH_FACES);
    CARVE(J.forig, int32_t, nfi + 1, PH_FACES, PH_CT); CARVE(J.s_of_o, int32_t, nfi + 1, PH_FACES, PH_WALK);
  }
  if (J.seq) {                                              // sequential connectivity: hash table, per-corner maps, the index section
    J.sq_cap = pow2_at_least(2ull * nc + 2);
    CARVE(J.sq_keys, unsigned long long, J.sq_cap, PH_DEDUP, PH_LAYOUT); CARVE(J.sq_val, uint32_t, J.sq_cap, PH_DEDUP, PH_LAYOUT);
    CARVE(J.sq_pu, int32_t, nc + 3, PH_DEDUP, PH_LAYOUT); CARVE(J.sq_first, int32_t, nc + 3, PH_DEDUP, PH_LAYOUT); CARVE(J.sq_pid, int32_t, nc + 3, PH_DEDUP, PH_LAYOUT);
    CARVE(J.sq_cop, int32_t, ecap + 1, PH_DEDUP, PH_LAYOUT); CARVE(J.sq_flag, uint8_t, nc + 3, PH_DEDUP, PH_LAYOUT); CARVE(J.sq_idx, uint8_t, 4 * nc + 16, PH_DEDUP, PH_LAYOUT);
  }
  // ---- pinned, zero-initialised ----
  CARVE(J.he_start, uint32_t, (size_t)J.n_pos + 1, PH_PINNED, PH_PINNED);
  CARVE(J.vvis, uint8_t, ecap / 8 + 64, PH_PINNED, PH_PINNED);
  CARVE(J.nmbits, uint32_t, (size_t)J.n_pos / 32 + 2, PH_PINNED, PH_PINNED);
  for (int i = 0; i < 2; i++) CARVE(J.vseam[i], uint32_t, ecap / 32 + 2, PH_PINNED, PH_PINNED);      // one bit per vertex: the whole map stays in L2
  for (int t = 0; t < 3; t++) CARVE(J.t_vvis[t], uint8_t, ecap / 8 + 64, PH_PINNED, PH_PINNED);
  for (int s = 0; s < GEO_NSTREAM; s++) {
    const int q = s == 6 ? J.qp : (s == 7 ? J.qt : J.qn);
    J.rs[s].alpha_cap = s < 6 ? 8 : (1u << (q + 1)) + 8;
    CARVE(J.rs[s].freq, uint32_t, J.rs[s].alpha_cap, PH_PINNED, PH_PINNED);
  }
  // ---- scan scratch (tiny, kept for the whole batch) ----
  CARVE(J.bsum, uint32_t, nc / UVOL_BLOCK + 8, PH_DEDUP, PH_LAYOUT); CARVE(J.bsum2, uint32_t, nc / UVOL_BLOCK + 8, PH_DEDUP, PH_LAYOUT);
  // ---- K2 / K3 ----
  { const int cl = J.seq ? PH_LAYOUT : PH_FACES;          // the sequential path reads the canonical ids when it quantises the points
    CARVE(J.canon[0], uint32_t, J.n_pos + 1, PH_DEDUP, cl); CARVE(J.canon[1], uint32_t, J.n_uv + 1, PH_DEDUP, cl); CARVE(J.canon[2], uint32_t, J.n_nrm + 1, PH_DEDUP, cl); }
  CARVE(J.keep, uint8_t, nfi + 1, PH_FACES, PH_FACES);
  // the stored corner table (canonical value ids, opposite corners, vertex ids) lives until the predictors: nothing is renumbered
  // (the compact layout has no copies: every frame's ids are its caller's index arrays, or the group is laid out again - geo_submit_impl)
  if (!J.compact) { CARVE(J.cp, int32_t, nc + 3, PH_FACES, PH_PRED); CARVE(J.cu, int32_t, nc + 3, PH_FACES, PH_PRED); CARVE(J.cn, int32_t, nc + 3, PH_FACES, PH_PRED); }
  CARVE(J.he_cur, uint32_t, (size_t)J.n_pos + 1, PH_CT, PH_FANS0); CARVE(J.he_ent, unsigned long long, nc + 1, PH_CT, PH_FANS0);      // k_vert0 walks the buckets
  { // partitioned bucket build (compact layout, ranges of <= HE_MAXVPB vertices): records + counts matrix, live in PH_CT only
    uint32_t vpb = 512; while ((uint64_t)vpb * HE_MAXBINS < (uint64_t)J.n_pos) vpb *= 2;
    J.he_vpb = (!full && vpb <= HE_MAXVPB && J.n_pos > 0) ? vpb : 0u;
    J.he_nb = J.he_vpb ? (J.n_pos + vpb - 1) / vpb : 0u; J.he_nblk = J.he_vpb ? (uint32_t)((nc + HE_TILE - 1) / HE_TILE) : 0u;
    if (J.he_vpb) { CARVE(J.he_part, uint32_t, 3 * nc + 4, PH_CT, PH_CT); CARVE(J.he_cnt, uint32_t, (size_t)J.he_nb * J.he_nblk + 2, PH_CT, PH_CT); }
  }
  const int aux_last = J.late_join ? PH_PRED : PH_SEAMS;            // what the auxiliary stream (valence replay, context scatter) reads lives until its join
  CARVE(J.opp, int32_t, nc + 3, PH_CT, PH_PRED);
  CARVE(J.vert, int32_t, nc + 3, PH_FANS0, PH_PRED);
  // ---- K4 ----
  auto rec_size = [&](int fmt) { return (size_t)(fmt == 2 ? 16 : (fmt == 1 ? 32 : 64)) * (nfi + 1); };
  const size_t rec_bytes = rec_size(fmtT);
  // With one record per face in both the walk and the traversals (formats 2 / 2) the base table of the traversals IS the walk's table
  // (J.base_hi, rec[1] = rec[0]: set after the placement); otherwise table 1 is a copy in the traversals' format (k_pack_tabs).
  const bool base_shared = fmt0 == 2 && fmtT == 2;
  // pending corners of the walkers on per-face records: forks that still wait for their left side - tens on a regular mesh; worst case one per face
  J.stcap = (uint32_t)(full ? nfi + 2 : nfi / 8 + 1024);
  CARVE(J.rec[0], uint8_t, rec_size(fmt0), PH_DENSE0, base_shared ? PH_V2D : PH_WALK); CARVE(J.vopen_d[0], uint8_t, ecap, PH_FANS0, PH_DENSE1);
  for (int w = base_shared ? 2 : 1; w < 4; w++) CARVE(J.rec[w], uint8_t, rec_bytes, PH_DENSE1, PH_V2D);
  CARVE(J.ring_d, int32_t, ecap, PH_FANS0, PH_SEAMS);
  CARVE(J.face_time, int32_t, nfi + 1, PH_DENSE0, std::max<int>(aux_last, PH_SEAMS));
  CARVE(J.proc, int32_t, nfi + 1, PH_WALK, aux_last); CARVE(J.symb, uint8_t, nfi + 64, PH_WALK, aux_last);
  CARVE(J.tstart, int32_t, nfi + 1, PH_FTIME, PH_TRAV);
  CARVE(J.initc, int32_t, nfi + 1, PH_WALK, PH_FTIME); CARVE(J.stack, int32_t, fmt0 == 2 ? J.stcap + 2 : nfi + 2, PH_WALK, PH_WALK); CARVE(J.start_bits, uint8_t, nfi + 1, PH_WALK, PH_ENT);
  // auxiliary stream (forked after PH_FTIME, joined before PH_HIST)
  CARVE(J.evcnt, uint8_t, nfi + 1, PH_RENUM, PH_SEAMS);
  // topology-split events: two per S symbol, a handful per mesh; worst case (retries) two per face
  J.evcap = (uint32_t)(full ? 2 * nfi + 2 : nfi / 8 + 256);
  CARVE(J.ev_src, int32_t, J.evcap, PH_RENUM, PH_LAYOUT); CARVE(J.ev_spl, int32_t, J.evcap, PH_RENUM, PH_LAYOUT); CARVE(J.ev_edge, uint8_t, J.evcap, PH_RENUM, PH_LAYOUT);
  CARVE(J.vval, int32_t, ecap + nfi + 3, PH_RENUM, aux_last); CARVE(J.c2vm, int32_t, nc + 3, PH_RENUM, aux_last); CARVE(J.ctx_of, uint8_t, nfi + 64, PH_RENUM, aux_last);
  CARVE(J.ctx_all, uint32_t, nfi + 8, PH_RENUM, PH_ENT);
  // ---- seams (on the stored tables: the decoder's order is virtual, GeoJob::tstart) ----
  CARVE(J.fseam, uint8_t, nfi + 1, PH_RENUM, PH_PRED); CARVE(J.sbpack, uint8_t, nfi + 1, PH_RENUM, PH_SEAMS);
  for (int i = 0; i < 2; i++) {
    CARVE(J.seam_bits[i], uint8_t, nc + 3, PH_SEAMS, PH_ENT);
    CARVE(J.avert[i], int32_t, nc + 3, PH_SEAMS, PH_PRED);          // (address space only: written for the corners of seam-touched vertices)
  }
  // ---- K5, K1, K6 ----
  for (int t = 0; t < 3; t++) { CARVE(J.order[t], int32_t, ecap + 4, PH_TRAV, PH_PRED); CARVE(J.v2d[t], int32_t, ecap, PH_V2D, PH_PRED); CARVE(J.t_stack[t], int32_t, fmtT == 2 ? J.stcap + 2 : nfi + 2, PH_TRAV, PH_TRAV); }
  if (J.seq) { CARVE(J.P, int32_t, 3 * ecap, PH_QUANT, PH_PRED); CARVE(J.U, int32_t, 2 * ecap, PH_QUANT, PH_PRED); CARVE(J.O, int32_t, 2 * ecap, PH_QUANT, PH_PRED); }
  else {                                                  // quantised values by value id (k_quant_ids), gathered by k_v2d and the predictors
    CARVE(J.qpos, uint16_t, 4 * ((size_t)J.n_pos + 1), PH_FACES, PH_PRED); CARVE(J.quv, uint16_t, 2 * ((size_t)J.n_uv + 1), PH_FACES, PH_PRED); CARVE(J.qnrm, uint16_t, 2 * ((size_t)J.n_nrm + 1), PH_FACES, PH_PRED);
  }
  CARVE(J.fnorm, int32_t, J.n_nrm ? (J.qp <= 15 ? 3 : 6) * nfi + 6 : 2, PH_PRED, PH_PRED);
  CARVE(J.sym_pos, uint32_t, 3 * ecap, PH_PRED, PH_ENT); CARVE(J.sym_uv, uint32_t, 2 * ecap, PH_PRED, PH_ENT); CARVE(J.sym_nrm, uint32_t, 2 * ecap, PH_PRED, PH_ENT);
  CARVE(J.has_ori, uint8_t, ecap, PH_PRED, PH_PRED); CARVE(J.ori_val, uint8_t, ecap, PH_PRED, PH_PRED); CARVE(J.ori_c, uint8_t, ecap, PH_PRED, PH_PRED);
  CARVE(J.ori_bits, uint8_t, ecap, PH_PRED, PH_ENT); CARVE(J.flips, uint8_t, ecap, PH_PRED, PH_ENT);
  // ---- K7 ----
  for (int s = 0; s < GEO_NSTREAM; s++) {
    RansStream &S = J.rs[s];
    const size_t nsym = s < 6 ? nfi : (s == 6 ? 3 * ecap : 2 * ecap);
    CARVE(S.probs, uint32_t, S.alpha_cap, PH_HIST, PH_LAYOUT); CARVE(S.cum, uint32_t, S.alpha_cap, PH_HIST, PH_LAYOUT);
    CARVE(S.head, uint8_t, 3 * (size_t)S.alpha_cap + 32, PH_HIST, PH_LAYOUT);
    CARVE(S.tab, uint4, S.alpha_cap, PH_HIST, PH_ENT);
    S.pay_cap = (uint32_t)(3 * nsym + 256); CARVE(S.pay, uint8_t, S.pay_cap, PH_ENT, PH_LAYOUT);
    // counting-sort scratch of k_rans_tables: (precision + 2) counters + one slot per symbol of the alphabet
    const int bl = bitlen(S.alpha_cap), pb = std::min(20, std::max(12, 3 * bl / 2));
    CARVE(S.scratch, uint32_t, ((size_t)1 << pb) + 4 + S.alpha_cap, PH_HIST, PH_HIST);
    S.syms = nullptr; S.n = 0; S.max_sym = 0; S.head_len = 0; S.pay_len = 0; S.pay_off = 0; S.prec_bits = 12;
  }
  for (int b = 0; b < GEO_NRABS; b++) {
    RabsStream &B = J.rb[b];
    const size_t nb = b == 0 ? nfi : (b < 3 ? nc : ecap);
    B.cap = (uint32_t)(nb / 4 + nb / 8 + 256); CARVE(B.buf, uint8_t, B.cap, PH_ENT, PH_LAYOUT);
    B.bits = nullptr; B.n = 0; B.zeros = 0; B.off = 0; B.len = 0;
  }
  J.arena_cap = (uint32_t)(20 * nfi + 1024); CARVE(J.arena, uint8_t, J.arena_cap, PH_LAYOUT, PH_LAYOUT);
#undef CARVE
}

// first-fit placement over the lifetime intervals; pinned items first (they form the zeroed head)
void ws_place(std::vector<WsItem> &items, WsPlan &P) {
  std::vector<UvolWsItem> w(items.size());
  for (size_t i = 0; i < items.size(); i++) w[i] = UvolWsItem{ items[i].bytes, items[i].first, items[i].last, 0 };
  P.total = uvol_ws_place(w, &P.zero, PH_LAYOUT + 1, "geometry encode");
  P.offs.resize(items.size());
  for (size_t i = 0; i < items.size(); i++) { items[i].off = w[i].off; P.offs[i] = w[i].off; }
}

// Lays out one job's workspace (sizes + capacities always; pointers when base != nullptr).  The capacities stored in J come from its
// own counts; the PLACEMENT is the one of the frame's shape bucket - its counts rounded up to multiples of 1024 (2048 faces), at most
// 1 % more bytes at 100 k vertices - so that the differing frames of a sequence share a handful of cached placements.
const WsPlan &layout_job(GeoJob &J, uint8_t *base, bool full, int fmt0, int fmtT, WsPlanCache &C, std::vector<WsItem> &items) {
  ws_collect(J, full, fmt0, fmtT, items);
  auto up = [](uint32_t v, uint32_t q) { return (uint64_t)((v + (uint64_t)q - 1) / q) * q; };
  const uint64_t flags = (uint64_t)J.qp | ((uint64_t)J.qt << 8) | ((uint64_t)J.qn << 16) | ((uint64_t)full << 24) | ((uint64_t)(J.relabel != 0) << 26) |
                         ((uint64_t)(J.seq != 0) << 27) | ((uint64_t)(J.late_join != 0) << 28) | ((uint64_t)fmt0 << 29) | ((uint64_t)fmtT << 31) | ((uint64_t)(J.compact != 0) << 33);      // everything ws_collect's sizes AND lifetimes depend on
  std::vector<uint64_t> key = { up(J.nf_in, 2048), up(J.n_pos, 1024), up(J.n_uv, 1024), up(J.n_nrm, 1024), flags, items.size(), 0 };
  auto it = C.plans.find(key);
  if (it == C.plans.end()) {
    GeoJob R = J; R.nf_in = (uint32_t)std::min<uint64_t>(key[0], 1u << 26); R.n_pos = (uint32_t)key[1]; R.n_uv = (uint32_t)key[2]; R.n_nrm = (uint32_t)key[3];
    std::vector<WsItem> ri; ws_collect(R, full, fmt0, fmtT, ri);
    bool ok = ri.size() == items.size();
    for (size_t i = 0; ok && i < ri.size(); i++) ok = ri[i].slot == items[i].slot && ri[i].bytes >= items[i].bytes && ri[i].first == items[i].first && ri[i].last == items[i].last;
    if (!ok) {                                             // a count on a structural boundary (an array the rounded shape does not have): exact placement for this shape
      key = { J.nf_in, J.n_pos, J.n_uv, J.n_nrm, flags, items.size(), 1 };
      it = C.plans.find(key);
      ri = items;
    }
    if (it == C.plans.end()) {
      if (C.plans.size() >= 512) C.plans.clear();
      WsPlan P; ws_place(ri, P); P.key = key;
      it = C.plans.emplace(key, std::move(P)).first;
    }
  }
  const WsPlan &P = it->second;
  if (base) for (size_t i = 0; i < items.size(); i++) *reinterpret_cast<uint8_t **>((char *)&J + items[i].slot) = base + P.offs[i];
  return P;
}
}  // namespace

// Upper bound of a .drc for quantisation bits <= 16: per value <= 2.5 bytes of rANS payload (20-bit precision) + 3 bytes of
// probability table per distinct symbol, 21 values per face at most; connectivity symbols, seam / orientation bits and the
// split events stay below 20 bytes per face; fixed headers and the zero runs of three 2^17-symbol tables below 64 KiB.
size_t uvol_mesh_bound(const uvol_mesh *m) {
  if (!m) return 0;
  return 65536 + (size_t)m->n_faces * 144;
}

#define LAUNCH(k, grid, block, ...)                                                              \
  do {                                                                                           \
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch %s\n", #k); fflush(stderr); }              \
    hipLaunchKernelGGL(k, grid, block, 0, ctx->stream, __VA_ARGS__);                             \
    if (uvol_debug()) { hipError_t e_ = hipStreamSynchronize(ctx->stream); if (e_ != hipSuccess) { fprintf(stderr, "[uvol] %s FAILED: %s\n", #k, hipGetErrorString(e_)); fflush(stderr); } \
      GeoJob dbg_; (void)hipMemcpy(&dbg_, dj, sizeof(GeoJob), hipMemcpyDeviceToHost); \
      fprintf(stderr, "[uvol]   job0 status %d nf %u nverts %u ne %u %u %u ne_uv %u has_ori %p bsum %p sbpack %p n_ori %u\n", dbg_.status, dbg_.nf, dbg_.nverts, dbg_.ne[0], dbg_.ne[1], dbg_.ne[2], dbg_.ne_uv, (void*)dbg_.has_ori, (void*)dbg_.bsum, (void*)dbg_.sbpack, dbg_.n_ori); fflush(stderr); } \
  } while (0)

#define LAUNCH_ON(stream_, k, grid, block, ...)                                                  \
  do {                                                                                           \
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch %s (aux)\n", #k); fflush(stderr); }        \
    hipLaunchKernelGGL(k, grid, block, 0, stream_, __VA_ARGS__);                                 \
    if (uvol_debug()) { hipError_t e_ = hipStreamSynchronize(stream_); if (e_ != hipSuccess) { fprintf(stderr, "[uvol] %s FAILED: %s\n", #k, hipGetErrorString(e_)); fflush(stderr); } } \
  } while (0)
#define LAUNCH_SM(k, grid, block, shmem, ...)                                                    \
  do {                                                                                           \
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch %s (lds %zu)\n", #k, (size_t)(shmem)); fflush(stderr); } \
    hipLaunchKernelGGL(k, grid, block, shmem, ctx->stream, __VA_ARGS__);                         \
    if (uvol_debug()) { hipError_t e_ = hipStreamSynchronize(ctx->stream); if (e_ != hipSuccess) { fprintf(stderr, "[uvol] %s FAILED: %s\n", #k, hipGetErrorString(e_)); fflush(stderr); } } \
  } while (0)

// zero the dedup hash tables / half-edge counts / visited maps / histograms at the head of every job's workspace
__global__ void __launch_bounds__(UVOL_BLOCK) k_job_clear(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  uint4 *p = reinterpret_cast<uint4 *>(J.ws_base);
  const size_t n16 = (size_t)(J.ws_zero / 16);
  for (size_t i = (size_t)blockIdx.x * UVOL_BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * UVOL_BLOCK) p[i] = make_uint4(0, 0, 0, 0);
}

// zero the three dedup hash tables of the jobs of one group (grid y = job, z = table)
__global__ void __launch_bounds__(UVOL_BLOCK) k_dd_clear(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.y];
  uint4 *p = reinterpret_cast<uint4 *>(J.dd_tab[blockIdx.z]);
  const size_t n16 = (size_t)J.dd_cap[blockIdx.z] / 4;                   // capacities are powers of two >= 4, tables 256-byte aligned
  for (size_t i = (size_t)blockIdx.x * UVOL_BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * UVOL_BLOCK) p[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && threadIdx.x == 0) for (uint32_t i = (uint32_t)n16 * 4; i < J.dd_cap[blockIdx.z]; i++) J.dd_tab[blockIdx.z][i] = 0;
}

// How the serial walkers of a batch run.  Wave-per-walker (one lane of a wave per walker, visited bitmaps in LDS) is the
// faster form per walker (one dependent load per face, 0.35 - 0.5 us) but a CU's LDS holds 3 of them; lane-per-walker (SIMT,
// nothing in LDS, 0.55 - 0.65 us per face at 16 lanes per wave) has no such cap.  So: LDS walkers while all of a launch's
// walkers are resident at once, SIMT walkers beyond that (measured on 720 frames in one launch: traversals 265 -> 135 ms).
// UVOL_SIMT_W=<1..64> forces the SIMT form with that many lanes per wave, UVOL_WALK_FORCE=global its one-lane-per-wave form
// (what a mesh too large for LDS gets), UVOL_WALK_FORCE=vglobal LDS walkers with their vertex bitmap in global memory (tests).
struct WalkPlan { int simt_w; size_t lds; int vcw; };
// locality relabelling: 2 = per frame, decided on the device (k_coherence: frames stored coherently skip it); UVOL_RELABEL=1 / 0 (tests, diagnostic) force it on / off
static inline int geo_relabel_mode() { static const int v = [] { const char *e = getenv("UVOL_RELABEL"); return !e ? 2 : (*e == '0' ? 0 : (*e == '1' ? 1 : 2)); }(); return v; }
static inline bool geo_relabel_on() { return geo_relabel_mode() != 0; }
static inline int geo_walk_pf() { static const int v = [] { const char *e = getenv("UVOL_WALK_PF"); return (e && *e == '0') ? 0 : 1; }(); return v; }     // UVOL_WALK_PF=0 (diagnostic): LDS walkers without their prefetch wave
static inline int geo_simt_env() { static const int w = [] { const char *e = getenv("UVOL_SIMT_W"); int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }(); return w; }
static WalkPlan walk_plan(const GeoState *G, uint32_t max_nfi, uint32_t max_vals, size_t n_walkers, bool vertex_bits_global = false) {
  WalkPlan P{0, 0, 0};
  const size_t walk_fw = ((size_t)max_nfi + 31) / 32;
  // Vertex bitmap capacity.  At least the largest attribute array of the batch + 6 % (vertices split at seams and
  // non-manifold fans; a table that still exceeds it keeps its vertex bitmap in global memory); then rounded UP to
  // whatever fits the same number of walkers per CU, so the slack of the LDS slot is not wasted.
  const size_t lds_cu = 150 * 1024 /* what several workgroups can share of a CU's 160 KiB (measured: 3 x 53 KiB does not fit) */, fw_bytes = walk_fw * 4 + (WALK_STG_DWORDS + WALK_PUB_DWORDS) * 4 + 16 /* + output staging, position word */;
  const size_t v_min_bytes = std::min<size_t>((((size_t)max_vals + max_vals / 16 + 31) / 32 + 2) * 4, ((3 * (size_t)max_nfi + 31) / 32) * 4);
  static const int walk_force = [] { const char *e = getenv("UVOL_WALK_FORCE"); return !e ? 0 : (!strcmp(e, "vglobal") ? 1 : (!strcmp(e, "global") ? 2 : 0)); }();
  const bool vglobal = walk_force == 1 || vertex_bits_global;
  size_t per_cu = lds_cu / (fw_bytes + (vglobal ? 4 : v_min_bytes)); if (per_cu < 1) per_cu = 1;
  const size_t slot = (lds_cu / per_cu) & ~(size_t)1023;
  const size_t walk_vcw = vglobal ? 1 : (slot > fw_bytes + v_min_bytes ? (slot - fw_bytes) / 4 : v_min_bytes / 4);
  P.lds = (((walk_fw + walk_vcw + 3) & ~(size_t)3) + WALK_STG_DWORDS + WALK_PUB_DWORDS) * 4; P.vcw = (int)walk_vcw;
  const bool lds_fits = P.lds <= G->max_lds;
  if (geo_simt_env() > 0) P.simt_w = geo_simt_env();
  else if (walk_force == 2 || !lds_fits) P.simt_w = 1;
  else if (walk_force == 0 && n_walkers > per_cu * (size_t)G->num_cu) P.simt_w = n_walkers >= 1024 ? 16 : 4;
  return P;
}
// 8-byte corner records (RecOps<true>): every corner code (< 4 * faces, signed field: 20 bits + sign) and every
// vertex id << 1 | open (unsigned 21-bit field) of the batch must fit: faces < 2^18 AND the vertex id space below 2^20.  Ids are
// position-based (n_pos + extra fans + seam segments, bounded by the workspace's entry capacity `ecap`), NOT bounded by the
// face count: a 2000-face mesh that references position 2^20 + 5 needs the 16-byte format.  UVOL_REC16=1 (tests) forces it.
static inline bool geo_rec8(uint32_t max_nfi, uint64_t max_ids) {
  static const bool force16 = [] { const char *e = getenv("UVOL_REC16"); return e && *e == '1'; }();
  return !force16 && 4ull * max_nfi < (1ull << 20) && max_ids < (1ull << 20);
}
// the decode path sizes its record tables with the same rule; its vertex ids are dense (< 3 * faces)
bool geo_records8(uint32_t max_nfi) { return geo_rec8(max_nfi, 3ull * max_nfi); }
static inline bool geo_rec_face_off() { static const bool v = [] { const char *e = getenv("UVOL_REC_FACE"); return e && *e == '0'; }(); return v; }      // UVOL_REC_FACE=0 (diagnostic, tests): corner records in the lane-per-walker kernels too
static void launch_traversals(uvol_ctx *ctx, GeoJob *dj, int n, const WalkPlan &P, int r8, int base_hi = 0) {
  const unsigned N = (unsigned)n;
  if (P.simt_w) {
    const unsigned W = (unsigned)P.simt_w, nb = (3 * N + W - 1) / W;
    // UVOL_TRAV_FORM=lane (tests, diagnostic): the lane-per-walker kernel whose idle lanes leave the loop; default: the wave form
    // (k_traverse_wave_f16: pops as ordinary steps, component search by the whole wave), UVOL_TRAV_W lanes per wave
    static const bool lane_form = [] { const char *e = getenv("UVOL_TRAV_FORM"); return e && !strcmp(e, "lane"); }();
    static const int wave_w = [] { const char *e = getenv("UVOL_TRAV_W"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();
    if (r8 == 2 && !lane_form) { const unsigned Ww = wave_w ? (unsigned)wave_w : W; LAUNCH(k_traverse_wave_f16, dim3((N + Ww - 1) / Ww, 3), dim3(64), dj, n, (int)Ww, 0, base_hi); }
    else if (r8 == 2) LAUNCH(k_traverse_simt_f16, dim3(nb), dim3(64), dj, n, (int)W, 0, 3);
    else if (r8) LAUNCH((k_traverse_simt<true>), dim3(nb), dim3(64), dj, n, (int)W); else LAUNCH((k_traverse_simt<false>), dim3(nb), dim3(64), dj, n, (int)W);
  }
  else if (r8) LAUNCH_SM((k_traverse<true>), dim3(3, N), dim3(128), P.lds, dj, P.vcw, geo_walk_pf() ? 2 : 0);
  else LAUNCH_SM((k_traverse<false>), dim3(3, N), dim3(128), P.lds, dj, P.vcw, geo_walk_pf() ? 2 : 0);
}
// attribute sequencing of a prepared GeoJob array (tables 1..3): corner records, DepthFirstTraverser, inverse maps.
// Shared with the decode path (geom_decode.hip), which fills the same job fields from a decoded corner table.
int geo_run_traversals(uvol_ctx *ctx, GeoJob *dj, int n, uint32_t max_nfi, uint32_t max_vals) {
  GeoState *G = ctx->geo;
  const unsigned N = (unsigned)n, bf = uvol_blocks(max_nfi), bc = uvol_blocks((size_t)3 * max_nfi);
  WalkPlan P = walk_plan(G, max_nfi, max_vals, (size_t)3 * N);
  // files of a batch are unrelated meshes as far as the decoder knows: one traverser per wave (several per wave pay for each other's
  // rare paths and misses: 953 against 293 ms per 2560 distinct frames with 16 / 1), and one 16-byte record per face where the fields allow it
  const int r8 = geo_records8(max_nfi) ? ((P.simt_w && !geo_rec_face_off()) ? 2 : 1) : 0;
  if (P.simt_w > 1 && geo_simt_env() == 0) P.simt_w = r8 == 2 ? 4 : 1;      // (four per wave in the wave form on per-face records)
  for (int w = 1; w <= 3; w++) LAUNCH(k_pack_faces, dim3(bf, N), dim3(UVOL_BLOCK), dj, w, r8);
  launch_traversals(ctx, dj, n, P, r8);
  LAUNCH(k_v2d, dim3(bc, N, 3), dim3(UVOL_BLOCK), dj, r8);
  return UVOL_OK;
}

// device workspace one frame of these dimensions holds while it is in flight (compact layout + its share of the packed output area)
extern "C" size_t uvol_mesh_workspace(const uvol_ctx *ctx, const uvol_mesh *m) {
  if (!ctx || !m || !m->n_faces) return 0;
  GeoJob J{}; J.compact = 1;        // (a clean frame of a large batch: the compact layout; a group with unclean frames holds 36 bytes per face more)
  J.relabel = geo_relabel_mode(); J.n_pos = m->n_pos; J.nf_in = m->n_faces; J.n_uv = (m->uv && m->idx_uv) ? m->n_uv : 0; J.n_nrm = (m->nrm && m->idx_nrm) ? m->n_nrm : 0;
  J.qp = ctx->prm.q_position_attr; J.qt = ctx->prm.q_texture_attr; J.qn = ctx->prm.q_normal_attr;
  WsPlanCache P; std::vector<WsItem> items;
  const int fmt = geo_rec8(m->n_faces, geo_ecap(J.n_pos, J.n_uv, J.n_nrm, J.nf_in, false)) ? 2 : 0;       // what a frame of a large batch holds (lane-per-walker kernels, one record per face)
  // a small call (and the one-frame retry) runs the LDS walkers on four 8-byte-per-corner tables instead: the larger of the two placements
  const size_t big = layout_job(J, nullptr, false, fmt, fmt, P, items).total;
  WsPlanCache P1; std::vector<WsItem> items1; GeoJob J1 = J; J1.compact = 0;
  const int fmt1 = fmt ? 1 : 0;
  const size_t small = layout_job(J1, nullptr, false, fmt1, fmt1, P1, items1).total;
  return std::max(big, small) + 32768 + 8 * (size_t)m->n_faces + sizeof(GeoJob);
}
// stages of a batch with sequential connectivity, between k_minmax and the layout (all parallel; see k_sq_*)
static int geo_encode_sequential(uvol_ctx *ctx, GeoJob *dj, int n, bool full, uint32_t max_nfi, uint32_t max_vals, uint32_t max_ecap, uint64_t algo_in) {
  const unsigned N = (unsigned)n, bc = uvol_blocks((size_t)3 * max_nfi), bv = uvol_blocks(max_vals), be = uvol_blocks(std::min<size_t>(max_ecap, (size_t)3 * max_nfi));
  const uvol_params &prm = ctx->prm;
  {
    uvol_ctx::Scope sc(ctx, "geo.k2_dedup", algo_in);
    if (!full) {
      uint32_t slots = DD_SLOTS; { const char *e = getenv("UVOL_DD_SLOTS"); const int v = e ? atoi(e) : 0; if (v >= 4 && v <= DD_SLOTS && !(v & (v - 1))) slots = (uint32_t)v; }
      const unsigned bt = (unsigned)((max_vals + DD_TILE - 1) / DD_TILE), nbm = (unsigned)std::min<uint64_t>(DD_MAXBINS, pow2_at_least(std::max<uint64_t>(1, max_vals / 1024)));
      LAUNCH(k_dd_count, dim3(bt, N, 3), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_dd_scan, dim3(1, N, 3), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_dd_scatter, dim3(bt, N, 3), dim3(UVOL_BLOCK), dj);
      if (max_vals / std::max(1u, nbm) <= 1100u && slots >= 2048u) LAUNCH((k_dd_resolve<2048, 4>), dim3(nbm, N, 3), dim3(UVOL_BLOCK), dj, 2048u);
      else LAUNCH((k_dd_resolve<DD_SLOTS, 6>), dim3(nbm, N, 3), dim3(UVOL_BLOCK), dj, slots);
    } else {
      LAUNCH(k_dd_clear, dim3(16, N, 3), dim3(UVOL_BLOCK), dj);
      for (int ph = 0; ph < 2; ph++) { LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 0, ph); LAUNCH(k_dedup<2>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 1, ph); LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 2, ph); }
    }
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.seq_points", (uint64_t)3 * max_nfi * 12);
    for (int level = 0; level < 2; level++) {
      LAUNCH(k_sq_clear, dim3(256, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_sq_hash, dim3(bc, N), dim3(UVOL_BLOCK), dj, level, 0);
      LAUNCH(k_sq_hash, dim3(bc, N), dim3(UVOL_BLOCK), dj, level, 1);
    }
    for (int step = 0; step < 4; step++) {
      LAUNCH(k_sq_points, dim3(bc, N), dim3(UVOL_BLOCK), dj, step);
      if (step == 0 || step == 2) LAUNCH(k_scan_sums, dim3(1, N), dim3(UVOL_BLOCK), dj, (int)SCAN_SEQ);
    }
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.k1_quantize", algo_in);
    LAUNCH(k_sq_quant, dim3(be, N, 3), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_sq_pred, dim3(be, N, 3), dim3(UVOL_BLOCK), dj);
  }
  {
    uvol_ctx::Scope sc0(ctx, "geo.k7_hist_tables", 0);
    LAUNCH(k_hist, dim3(uvol_blocks((size_t)9 * max_nfi, 16 * UVOL_BLOCK), GEO_NSTREAM, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_rans_tables, dim3(GEO_NSTREAM, N), dim3(64), dj);
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.k7_entropy_encode", 0);
    unsigned W = 4; while (W < 64 && 5u * N > 512u * W) W *= 2;
    LAUNCH(k_rans_recip, dim3(uvol_blocks(((size_t)2 << std::max(prm.q_position_attr, std::max(prm.q_texture_attr, prm.q_normal_attr))) + 8), GEO_NSTREAM, N), dim3(UVOL_BLOCK), dj);
    LAUNCH_SM(k_entropy_simt, dim3((N + W - 1) / W, GEO_NSTREAM), dim3(64), (size_t)W * SB_STRIDE * 4, dj, n, (int)W);
  }
  return UVOL_OK;
}
static int geo_submit(uvol_ctx *ctx, GeoLane &L, const uvol_mesh *meshes, int n, int n_conc, bool on_device,
                      uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status, bool full, GeoUp *pre = nullptr);
static int geo_complete(uvol_ctx *ctx, GeoLane &L);

// First half of a group of frames on lane L: lays out the workspaces, uploads host inputs, enqueues every kernel of the group and the
// read-back of its job records.  Returns without waiting for the GPU (but for the one look at the batch's storage order, below).
// full = worst-case workspace and output sizes (the retry of a frame the compact layout could not hold); n_conc = frames of the whole
// call (what is on the chip together decides the kernel forms, not this group's share).
static int geo_submit_impl(uvol_ctx *ctx, GeoLane &L, const uvol_mesh *meshes, int n, int n_conc, bool on_device, const size_t *caps, bool full) {
  GeoState *G = ctx->geo;
  const auto t_enter = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  const uvol_params &prm = ctx->prm;
  if (prm.q_position_attr < 1 || prm.q_position_attr > 16 || prm.q_texture_attr < 1 || prm.q_texture_attr > 16 ||
      prm.q_normal_attr < 2 || prm.q_normal_attr > 16) { ctx->set_error("quantization bits out of supported range (1..16)"); return UVOL_E_UNSUPPORTED; }
  // DRACO_COMPRESSION_LEVEL only selects the encoder's tools, it is not written to the file (scripts/Encoder.py:171-179 documents
  // 0..10): every legal level is encoded with the cl-7 tool set (edgebreaker + valence contexts, parallelogram / tex-coord /
  // geometric-normal prediction, RAW rANS), which any Draco decoder reads; the shims say so on stderr
  if (prm.draco_compression_level < 0 || prm.draco_compression_level > 10) { ctx->set_error("DRACO_COMPRESSION_LEVEL %d outside 0..10", prm.draco_compression_level); return UVOL_E_INVALID; }
  // DRACO_COMPRESSION_LEVEL 0 selects sequential connectivity in stock draco_encoder (speed 10); every other level is written with the
  // level-7 tool set (valence edgebreaker)
  const bool seq = prm.draco_compression_level == 0;
  // The valence replay is one wave per frame and takes what one frame takes (~25-50 ms); the renumber / seams group it runs beside shrinks
  // with the batch.  Below ~1200 frames the replay is the longer of the two, so it is joined late (before the entropy stage) and
  // overlaps the traversals too; its inputs then cannot share bytes with the record tables (+7.7 MB per frame, irrelevant at that size).
  static const int late_env = [] { const char *e = getenv("UVOL_LATE_JOIN"); return e ? atoi(e) : -1; }();      // tests: 0 / 1 force the early / late join
  const bool late_join = late_env >= 0 ? late_env != 0 : std::max(n, n_conc) <= 1200;      // (n_conc: the frames of the whole call are on the chip together, whatever this group's share)
  std::vector<size_t> ws_off(n), in_off(n), zero_sz(n);
  size_t ws_total = 0, in_total = 0, out_total = 0;
  uint32_t max_nfi = 0, max_vals = 0, max_ecap = 0, he_nb_max = 0, ms_nb_max = 1; bool he_part_all = true; uint64_t algo_in = 0;
  uint64_t max_ids = 0;
  for (int i = 0; i < n; i++) {
    const uvol_mesh &m = meshes[i]; max_nfi = std::max(max_nfi, m.n_faces);
    max_ids = std::max(max_ids, geo_ecap(m.n_pos, (m.uv && m.idx_uv) ? m.n_uv : 0, (m.nrm && m.idx_nrm) ? m.n_nrm : 0, m.n_faces, full));
    max_vals = std::max(max_vals, std::max(m.n_pos, std::max((m.uv && m.idx_uv) ? m.n_uv : 0u, (m.nrm && m.idx_nrm) ? m.n_nrm : 0u)));
  }
  const int r8 = geo_rec8(max_nfi, max_ids) ? 1 : 0;
  // How the serial walkers of this group run is decided here, before the workspaces are laid out: the lane-per-walker kernels read ONE
  // 16-byte record per face (format 2) where the 21-bit fields allow it, and the record tables are sized for the format
  const unsigned NC0 = (unsigned)std::max(n, n_conc);
  WalkPlan wp_walk = walk_plan(G, max_nfi, max_vals, (size_t)NC0);
  static const bool tvg_env = [] { const char *e = getenv("UVOL_TRAVERSE_VGLOBAL"); return e && *e == '1'; }();
  const bool tvg = tvg_env || ctx->prm.traverse_vbits_l2 != 0;
  WalkPlan wp_trav = walk_plan(G, max_nfi, max_vals, (size_t)3 * NC0, tvg);
  // (Measured in round 5 and not kept, profiles/r05_small_jobs.json: between 256 and 512 frames the LDS traversers with their VERTEX bitmap in L2
  // - 6 per CU, so 3 N of them fit - lose to the wave-form lane kernels: 190 against 124 ms per 300-frame job, 894 against 1110 frames/s.)
  const bool f16_off = geo_rec_face_off();
  const int fmt0 = (r8 && wp_walk.simt_w && !f16_off) ? 2 : r8, fmtT = (r8 && wp_trav.simt_w && !f16_off) ? 2 : r8;
  const bool base_shared = fmt0 == 2 && fmtT == 2;
  // COMPACT layout (round 5): a group is laid out without the arrays only an unclean or incoherently stored frame needs - the stored
  // copies of the canonical value ids (a clean frame's ids ARE the caller's index arrays, k_compact_faces) and the scratch of the
  // locality relabelling: 7.2 MB of every 200 k-face frame's workspace at its peak, i.e. more frames in flight.  Whether the frames are
  // clean is known after the dedup (k_coherence / k_relabel_decide count the others, read back with the relabel decision); if one is
  // not, the group is laid out again in the general form and starts over - the cost of one dedup pass, paid by groups that have such
  // frames, and the NEXT group starts in the general form at once (GeoState::compact_ok) until a general group turns out all clean.
  static const bool compact_env = [] { const char *e = getenv("UVOL_COMPACT_IDS"); return !(e && *e == '0'); }();
  static const int face_alias = [] { const char *e = getenv("UVOL_FACE_ALIAS"); return (e && *e == '0') ? 0 : 1; }();
  const bool can_probe = (geo_relabel_on() && !seq) || std::max(n, n_conc) >= 256;      // (the mid-batch read-back below happens)
  bool compact = seq || (compact_env && face_alias && !full && can_probe && G->compact_ok);
  int rc;
  bool uploaded = false;
  const bool pre_up = !on_device && L.up.slot != nullptr;
  auto lay = [&]() -> int {
  L.hjobs.assign((size_t)n, GeoJob{});
  ws_total = 0; in_total = 0; out_total = 0; max_ecap = 0; he_nb_max = 0; ms_nb_max = 1; he_part_all = true; algo_in = 0;
  for (int i = 0; i < n; i++) {
    const uvol_mesh &m = meshes[i]; GeoJob &J = L.hjobs[i];
    J.compact = compact ? 1 : 0;
    if (!m.pos || !m.idx_pos || m.n_pos == 0 || m.n_faces == 0 || m.n_faces > (1u << 26)) { ctx->set_error("mesh %d: empty or invalid", i); return UVOL_E_INVALID; }
    J.n_pos = m.n_pos; J.nf_in = m.n_faces; J.relabel = seq ? 0 : geo_relabel_mode(); J.seq = seq ? 1 : 0; J.late_join = late_join ? 1 : 0;
    J.has_uv = (m.uv && m.idx_uv && m.n_uv) ? 1 : 0; J.has_nrm = (m.nrm && m.idx_nrm && m.n_nrm) ? 1 : 0;
    J.n_uv = J.has_uv ? m.n_uv : 0; J.n_nrm = J.has_nrm ? m.n_nrm : 0;
    J.nad = J.has_uv + J.has_nrm; J.qp = prm.q_position_attr; J.qt = prm.q_texture_attr; J.qn = prm.q_normal_attr;
    { int k = 0; if (J.has_uv) J.att_kind[k++] = 0; if (J.has_nrm) J.att_kind[k++] = 1; for (; k < 2; k++) J.att_kind[k] = -1; }
    const WsPlan &wp = layout_job(J, nullptr, full, fmt0, fmtT, G->plan, G->items);
    ws_off[i] = ws_total; ws_total += wp.total; zero_sz[i] = wp.zero;
    const size_t in_sz = ((size_t)m.n_pos * 12 + 255) / 256 * 256 + ((size_t)J.n_uv * 8 + 255) / 256 * 256 + ((size_t)J.n_nrm * 12 + 255) / 256 * 256 +
                         (size_t)(1 + J.has_uv + J.has_nrm) * (((size_t)m.n_faces * 12 + 255) / 256 * 256);
    in_off[i] = in_total; in_total += (on_device || pre_up) ? 0 : in_sz;
    // the frames' streams are packed back to back (k_out_offsets), so the batch's output area is sized for typical streams
    // (8 bytes per face; the defaults give 1.3), not for the sum of the callers' capacities; GEO_E_SLAB_FULL -> retried alone
    const size_t oc = full ? caps[i] : std::min<size_t>(caps[i], 32768 + 8 * (size_t)m.n_faces);
    out_total += (oc + 255) & ~(size_t)255; J.out_cap = (uint32_t)std::min<size_t>(caps[i], 0xffffffffu);
    max_vals = std::max(max_vals, std::max(m.n_pos, std::max(J.n_uv, J.n_nrm))); max_ecap = std::max(max_ecap, J.ecap);
    he_nb_max = std::max(he_nb_max, J.he_nb); he_part_all = he_part_all && J.he_vpb != 0;
    if (J.relabel && !J.compact) ms_nb_max = std::max(ms_nb_max, std::max(J.ms_nb[0], ((J.n_pos ? J.n_pos - 1 : 0) >> J.ms_sh[1]) + 1));
    algo_in += (uint64_t)m.n_pos * 12 + (uint64_t)J.n_uv * 8 + (uint64_t)J.n_nrm * 12 + (uint64_t)(1 + J.has_uv + J.has_nrm) * m.n_faces * 12;
  }
  if ((rc = uvol_ensure(ctx, L.slab, ws_total))) {
    // out of device memory: the workspaces idle lanes still hold from earlier (larger) groups are given back, then once more
    (void)hipGetLastError();
    for (GeoLane *o : G->lanes) if (o != &L && o->busy) { const int r = geo_complete(ctx, *o); if (r != UVOL_OK && G->deferred_rc == UVOL_OK) G->deferred_rc = r; }      // (their results first)
    for (GeoLane *o : G->lanes) if (o != &L && !o->busy) for (uvol_devbuf *b : { &o->slab, &o->inputs, &o->outs }) if (b->p) { (void)hipStreamSynchronize(o->stream); (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    // ... and the ring gets by with one lane less from here on (until uvol_trim): otherwise every submission takes another lane's workspace away
    // and allocates its own again (the shuffled-order variant inside the bench line: 1657 frames/s against 2453 as a process of its own)
    G->lanes_cap = std::max(1, std::min(G->lanes_cap, (int)G->lanes.size()) - 1);
    if ((rc = uvol_ensure(ctx, L.slab, ws_total))) return rc;
  }
  if ((rc = uvol_ensure(ctx, L.jobs, sizeof(GeoJob) * (size_t)n))) return rc;
  if (!L.ext_out && (rc = uvol_ensure(ctx, L.outs, out_total))) return rc;
  if (!on_device && !pre_up && (rc = uvol_ensure(ctx, L.inputs, in_total))) return rc;
  std::vector<UvolUpItem> ups; if (!on_device && !pre_up) ups.reserve((size_t)n * 6);
  for (int i = 0; i < n; i++) {
    const uvol_mesh &m = meshes[i]; GeoJob &J = L.hjobs[i];
    uint8_t *base = (uint8_t *)L.slab.p + ws_off[i];
    (void)layout_job(J, base, full, fmt0, fmtT, G->plan, G->items);
    if (base_shared) { J.rec[1] = J.rec[0]; J.base_hi = 1; }      // the traversals' base table is the walk's record table (flag bit 127)
    J.ws_base = base; J.ws_zero = zero_sz[i];          // cleared by ONE k_job_clear launch for the whole batch (was 2 memsets per frame)
    if (L.ext_out) { J.out_pack = L.ext_out; J.slab_cap = L.ext_cap; } else { J.out_pack = (uint8_t *)L.outs.p; J.slab_cap = out_total; }
    if (on_device) { J.pos = m.pos; J.uv = J.has_uv ? m.uv : nullptr; J.nrm = J.has_nrm ? m.nrm : nullptr; J.ipos = m.idx_pos; J.iuv = J.has_uv ? m.idx_uv : nullptr; J.inrm = J.has_nrm ? m.idx_nrm : nullptr; }
    else if (pre_up) {                                       // already on their way: the group's uplink slot (geo_encode_batch_begin)
      const uint8_t *sb = (const uint8_t *)L.up.slot->buf.p; const size_t *o6 = &L.up.off[(size_t)i * 6];
      J.pos = (const float *)(sb + o6[0]); J.uv = J.has_uv ? (const float *)(sb + o6[1]) : nullptr; J.nrm = J.has_nrm ? (const float *)(sb + o6[2]) : nullptr;
      J.ipos = (const uint32_t *)(sb + o6[3]); J.iuv = J.has_uv ? (const uint32_t *)(sb + o6[4]) : nullptr; J.inrm = J.has_nrm ? (const uint32_t *)(sb + o6[5]) : nullptr;
    }
    else {
      uint8_t *ib = (uint8_t *)L.inputs.p + in_off[i]; size_t o = 0;
      auto up = [&](const void *src, size_t bytes) -> const void * {                 // queued: ONE staged upload for the whole batch below
        void *d = ib + o; if (bytes) ups.push_back(UvolUpItem{ in_off[i] + o, src, bytes }); o += (bytes + 255) / 256 * 256; return d; };
      J.pos = (const float *)up(m.pos, (size_t)m.n_pos * 12);
      J.uv = J.has_uv ? (const float *)up(m.uv, (size_t)m.n_uv * 8) : nullptr;
      J.nrm = J.has_nrm ? (const float *)up(m.nrm, (size_t)m.n_nrm * 12) : nullptr;
      J.ipos = (const uint32_t *)up(m.idx_pos, (size_t)m.n_faces * 12);
      J.iuv = J.has_uv ? (const uint32_t *)up(m.idx_uv, (size_t)m.n_faces * 12) : nullptr;
      J.inrm = J.has_nrm ? (const uint32_t *)up(m.idx_nrm, (size_t)m.n_faces * 12) : nullptr;
    }
    for (int k = 0; k < 3; k++) { J.pos_min_u[k] = 0xffffffffu; J.pos_max_u[k] = 0; }
    for (int k = 0; k < 2; k++) { J.uv_min_u[k] = 0xffffffffu; J.uv_max_u[k] = 0; J.wrap_lo[k] = 0x7fffffff; J.wrap_hi[k] = -0x7fffffff - 1; }
    // stream wiring
    for (int s = 0; s < 6; s++) J.rs[s].syms = J.ctx_sym[s];
    J.rs[6].syms = J.sym_pos; J.rs[7].syms = J.sym_uv; J.rs[8].syms = J.sym_nrm;
    J.rb[0].bits = J.start_bits; J.rb[1].bits = J.seam_bits[0]; J.rb[2].bits = J.seam_bits[1]; J.rb[3].bits = J.ori_bits; J.rb[4].bits = J.flips;
    // rabs slot 1/2 follow the attribute-data slot; slot 3 = uv orientations, slot 4 = normal flips
  }
  if (!on_device && !pre_up && !uploaded) { const int rcu = uvol_upload_staged(ctx, (uint8_t *)L.inputs.p, ups); if (rcu != UVOL_OK) return rcu; uploaded = true; }
  if (pre_up && !uploaded) { const int rcu = uvol_uplink_acquire(ctx, L.up.slot, ctx->stream); if (rcu != UVOL_OK) return rcu; uploaded = true; }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(L.jobs.p, L.hjobs.data(), sizeof(GeoJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  return UVOL_OK; };
  if ((rc = lay())) return rc;
  GeoJob *dj = (GeoJob *)L.jobs.p;
  const unsigned N = (unsigned)n, NC = (unsigned)std::max(n, n_conc);      // NC: frames on the chip together (all groups of the call)
  L.t_prep = ms_since(t_enter);
  if (L.producer) {                                        // the inputs are produced on the caller's stream: ordered after what it holds now, no host wait
    if (!L.ev_prod) UVOL_HIP_CHECK(ctx, hipEventCreateWithFlags(&L.ev_prod, hipEventDisableTiming));
    UVOL_HIP_CHECK(ctx, hipEventRecord(L.ev_prod, L.producer));
    UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, L.ev_prod, 0));
  }
  // The front ends (dedup, corner table: streaming kernels that fill the chip) of consecutive groups run one after the other, so that a
  // group's front end meets the WALKERS of the groups before it - latency-bound, a few hundred waves - instead of their front ends:
  // each group then gets through its bandwidth-bound phases at close to the chip's full rate and the groups stay staggered.
  // UVOL_GEO_CHAIN: 0 = no chain, 1 = behind the previous group's front end, 2 = behind the previous group's WALK (experiment: a stagger of
  // front end + walk, about a third of a group's chain, whatever the moment the host submitted the groups - three groups submitted together
  // otherwise stay bunched: their walkers run together, then their streaming kernels compete)
  static const int fe_chain = [] { const char *e = getenv("UVOL_GEO_CHAIN"); return e ? atoi(e) : 1; }();
  if (fe_chain == 2 && G->walk_last && G->walk_last != L.ev_walk) UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, G->walk_last, 0));
  else if (fe_chain && G->fe_last && G->fe_last != L.ev_fe) UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, G->fe_last, 0));
  const unsigned bf = uvol_blocks(max_nfi), bc = uvol_blocks((size_t)3 * max_nfi), bv = uvol_blocks(max_vals), bci = (bc + GEO_ILP - 1) / GEO_ILP;
  unsigned be = uvol_blocks(std::min<size_t>(max_ecap, (size_t)3 * max_nfi));       // attribute entries (<= ecap, else GEO_E_WS_OVERFLOW)
  const bool relabel = geo_relabel_on() && !seq;
  bool lockstep = true;                                      // walkers of this batch move in lock step (see below); decides lanes per wave of the traversers
  bool any_relabel = relabel;
  // UVOL_FE_SLICE=<frames> (experiment): the streaming front end (dedup; faces, corner table, the walk's record table)
  // is launched slice by slice of that many frames instead of kernel by kernel over the whole group; 0 / unset = whole group
  static const unsigned fe_slice_env = [] { const char *e = getenv("UVOL_FE_SLICE"); const int v = e ? atoi(e) : 0; return v <= 0 ? 0u : (unsigned)v; }();
  const unsigned fe_slice = fe_slice_env ? fe_slice_env : N;
  for (int attempt = 0;; attempt++) {
  LAUNCH(k_job_clear, dim3(128, N), dim3(UVOL_BLOCK), dj);
  // bounding boxes first: the relabelling's Morton keys are taken over them (k_quantize uses them much later)
  LAUNCH(k_minmax, dim3(std::min(bv, 16u), N), dim3(UVOL_BLOCK), dj);
  if (seq) break;
  {
    uvol_ctx::Scope sc(ctx, "geo.k2_dedup", algo_in);
    if (!full) {
      // UVOL_DD_SLOTS=<power of two <= 4096> (tests): LDS slots per bin; a small table forces GEO_E_DD_OVERFLOW and the retry
      uint32_t slots = DD_SLOTS; { const char *e = getenv("UVOL_DD_SLOTS"); const int v = e ? atoi(e) : 0; if (v >= 4 && v <= DD_SLOTS && !(v & (v - 1))) slots = (uint32_t)v; }
      const unsigned bt = (unsigned)((max_vals + DD_TILE - 1) / DD_TILE), nbm = (unsigned)std::min<uint64_t>(DD_MAXBINS, pow2_at_least(std::max<uint64_t>(1, max_vals / 1024)));
      for (unsigned s0 = 0; s0 < N; s0 += fe_slice) {          // (slices: see fe_slice above)
        GeoJob *djs = dj + s0; const unsigned Ns = std::min(fe_slice, N - s0);
        LAUNCH(k_dd_count, dim3(bt, Ns, 3), dim3(UVOL_BLOCK), djs);
        LAUNCH(k_dd_scan, dim3(1, Ns, 3), dim3(UVOL_BLOCK), djs);
        LAUNCH(k_dd_scatter, dim3(bt, Ns, 3), dim3(UVOL_BLOCK), djs);
        if (max_vals / std::max(1u, nbm) <= 1100u && slots >= 2048u) LAUNCH((k_dd_resolve<2048, 4>), dim3(nbm, Ns, 3), dim3(UVOL_BLOCK), djs, 2048u);
        else LAUNCH((k_dd_resolve<DD_SLOTS, 6>), dim3(nbm, Ns, 3), dim3(UVOL_BLOCK), djs, slots);
      }
    } else {
      LAUNCH(k_dd_clear, dim3(16, N, 3), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 0, 0);
      LAUNCH(k_dedup<2>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 1, 0);
      LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 2, 0);
      LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 0, 1);
      LAUNCH(k_dedup<2>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 1, 1);
      LAUNCH(k_dedup<3>, dim3(bv, N), dim3(UVOL_BLOCK), dj, 2, 1);
    }
    // How the frames are stored decides three things, so the batch is looked at once (k_coherence: one pass over the index arrays) and the
    // counts come back to the host (the only mid-batch synchronisation; the encode kernels of a large batch take 0.2 - 0.9 s):
    //  * frames stored coherently skip the locality relabelling - if none needs it, its ~6 M (empty) workgroups are not even launched;
    //  * frames with the SAME connectivity as their predecessor (an animated mesh of fixed topology) are walked in lock step by the
    //    lanes of a wave - 16 attribute traversers per wave then beat one per wave (200 vs 300 ms per 2160 frames), while walkers on
    //    unrelated meshes diverge and miss at different times, and fewer per wave are the faster form;
    //  * a group in the compact layout has to be all clean, coherently stored frames (above).
    any_relabel = relabel;
    if (relabel || NC >= 256) {
      if (!L.counts) UVOL_HIP_CHECK(ctx, hipMalloc((void **)&L.counts, 64));
      uint32_t hc[3] = { 0, 0, 0 };
      UVOL_HIP_CHECK(ctx, hipMemsetAsync(L.counts, 0, 64, ctx->stream));
      LAUNCH(k_coherence, dim3(bf, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_relabel_decide, dim3((N + 63) / 64), dim3(64), dj, n, L.counts);
      UVOL_HIP_CHECK(ctx, hipMemcpyAsync(hc, L.counts, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
      UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      any_relabel = relabel && hc[0] != 0;
      lockstep = (uint64_t)hc[1] * 10u >= (uint64_t)n * 9u;
      if (compact && hc[2] != 0 && attempt == 0) {           // a frame with duplicate values, a degenerate face or an incoherent storage order: general layout, once more
        compact = false; G->compact_ok = false;
        if ((rc = lay())) return rc;
        dj = (GeoJob *)L.jobs.p; be = uvol_blocks(std::min<size_t>(max_ecap, (size_t)3 * max_nfi));
        continue;
      }
      if (!compact && !full && hc[2] == 0) G->compact_ok = true;
    }
  }
  break;
  }
  if (seq) { const int rcq = geo_encode_sequential(ctx, dj, n, full, max_nfi, max_vals, max_ecap, algo_in); if (rcq != UVOL_OK) return rcq; UVOL_HIP_CHECK(ctx, hipEventRecord(L.ev_fe, ctx->stream)); G->fe_last = L.ev_fe; G->walk_last = nullptr; }
  else {
  const bool fe_sliced = fe_slice < N && !any_relabel && he_part_all;
  if (fe_sliced) {
    // the front end frame-slice by frame-slice: every kernel of the slice before the next slice starts, so that what one kernel writes
    // (canonical faces, edge records, buckets, opposite corners: ~30 MB per frame) is still in the 256 MiB Infinity Cache when the next reads it
    uvol_ctx::Scope sc(ctx, "geo.k3_corner_table", (uint64_t)3 * max_nfi * 4 * 3);
    const unsigned bt = (unsigned)(((size_t)3 * max_nfi + HE_TILE - 1) / HE_TILE);
    for (unsigned s0 = 0; s0 < N; s0 += fe_slice) {
      GeoJob *djs = dj + s0; const unsigned Ns = std::min(fe_slice, N - s0);
      LAUNCH(k_faces, dim3(bf, Ns), dim3(UVOL_BLOCK), djs);
      LAUNCH(k_scan_sums, dim3(1, Ns), dim3(UVOL_BLOCK), djs, (int)SCAN_KEEP);
      LAUNCH(k_compact_faces, dim3(bf, Ns), dim3(UVOL_BLOCK), djs, face_alias);
      LAUNCH(k_quant_ids, dim3(bv, Ns, 3), dim3(UVOL_BLOCK), djs);
      LAUNCH(k_hp_count, dim3(bt, Ns), dim3(UVOL_BLOCK), djs);
      LAUNCH(k_hp_scan, dim3(1, Ns), dim3(UVOL_BLOCK), djs);
      LAUNCH(k_hp_scatter, dim3(bt, Ns), dim3(UVOL_BLOCK), djs);
      LAUNCH(k_hp_build, dim3(he_nb_max, Ns), dim3(UVOL_BLOCK), djs);
      LAUNCH(k_edge_match, dim3(bci, Ns), dim3(UVOL_BLOCK), djs);
      LAUNCH(k_vert0, dim3(bv, Ns), dim3(UVOL_BLOCK), djs);
      LAUNCH(k_vert_fill, dim3(bc, Ns), dim3(UVOL_BLOCK), djs);
      LAUNCH(k_pack0, dim3(bf, Ns), dim3(UVOL_BLOCK), djs, fmt0);
    }
  } else {
  {
    uvol_ctx::Scope sc(ctx, "geo.k2b_faces", 0);
    const unsigned mt0 = (unsigned)((max_vals + MS_TILE - 1) / MS_TILE), mt1 = (unsigned)(((size_t)max_nfi + MS_TILE - 1) / MS_TILE);
    if (any_relabel) {                                       // new position ids (Morton order) and the positions in that order
      LAUNCH(k_ms_key_pos, dim3(bv, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_ms_count, dim3(mt0, N), dim3(UVOL_BLOCK), dj, 0);
      LAUNCH(k_ms_scan, dim3(1, N), dim3(UVOL_BLOCK), dj, 0);
      LAUNCH(k_ms_scatter, dim3(mt0, N), dim3(UVOL_BLOCK), dj, 0);
      LAUNCH(k_ms_place, dim3(ms_nb_max, N), dim3(UVOL_BLOCK), dj, 0);
    }
    LAUNCH(k_faces, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_scan_sums, dim3(1, N), dim3(UVOL_BLOCK), dj, (int)SCAN_KEEP);
    if (any_relabel) {                                       // faces stored in the order of their lowest new vertex id
      LAUNCH(k_face_cidx, dim3(bf, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_ms_count, dim3(mt1, N), dim3(UVOL_BLOCK), dj, 1);
      LAUNCH(k_ms_scan, dim3(1, N), dim3(UVOL_BLOCK), dj, 1);
      LAUNCH(k_ms_scatter, dim3(mt1, N), dim3(UVOL_BLOCK), dj, 1);
      LAUNCH(k_ms_place, dim3(ms_nb_max, N), dim3(UVOL_BLOCK), dj, 1);
      LAUNCH(k_relabel_faces, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    }
    LAUNCH(k_compact_faces, dim3(bf, N), dim3(UVOL_BLOCK), dj, face_alias);       // the frames that are not relabelled (decided per frame on the device)
    { uvol_ctx::Scope sq(ctx, "geo.k1_quantize", algo_in); LAUNCH(k_quant_ids, dim3(bv, N, 3), dim3(UVOL_BLOCK), dj); }      // (after the relabelling: position ids are the stored ones)
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.k3_corner_table", (uint64_t)n * 0 + (uint64_t)3 * max_nfi * 4 * 3);
    if (he_part_all) {
      const unsigned bt = (unsigned)(((size_t)3 * max_nfi + HE_TILE - 1) / HE_TILE);
      LAUNCH(k_hp_count, dim3(bt, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_hp_scan, dim3(1, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_hp_scatter, dim3(bt, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_hp_build, dim3(he_nb_max, N), dim3(UVOL_BLOCK), dj);
    } else {
      LAUNCH(k_he_count, dim3(bci, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_he_scan, dim3(1, N), dim3(UVOL_BLOCK), dj);
      LAUNCH(k_he_fill, dim3(bci, N), dim3(UVOL_BLOCK), dj);
    }
    LAUNCH(k_edge_match, dim3(bci, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_vert0, dim3(bv, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_vert_fill, dim3(bc, N), dim3(UVOL_BLOCK), dj);            // (frames with non-manifold vertices only: geo_vt)
  }
  }   // !fe_sliced
  UVOL_HIP_CHECK(ctx, hipEventRecord(L.ev_fe, ctx->stream)); G->fe_last = L.ev_fe;
  // UVOL_SIMT_W_WALK / UVOL_SIMT_W_TRAV (diagnostic): lanes per wave of one of the two lane-per-walker kernels only (UVOL_SIMT_W sets both)
  static const int w_walk_env = [] { const char *e = getenv("UVOL_SIMT_W_WALK"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();
  static const int w_trav_env = [] { const char *e = getenv("UVOL_SIMT_W_TRAV"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();
  if (w_walk_env && wp_walk.simt_w) wp_walk.simt_w = w_walk_env;
  {
    if (!fe_sliced) LAUNCH(k_pack0, dim3(bf, N), dim3(UVOL_BLOCK), dj, fmt0);
    uvol_ctx::Scope sc(ctx, "geo.k4_eb_walk", (uint64_t)n * 32 * max_nfi);
    if (wp_walk.simt_w) {
      const unsigned W = (unsigned)wp_walk.simt_w, nb = (N + W - 1) / W;
      if (fmt0 == 2) LAUNCH(k_eb_walk_simt_f16, dim3(nb), dim3(64), dj, n, (int)W);
      else if (r8) LAUNCH((k_eb_walk_simt<true>), dim3(nb), dim3(64), dj, n, (int)W); else LAUNCH((k_eb_walk_simt<false>), dim3(nb), dim3(64), dj, n, (int)W);
    }
    else if (r8) LAUNCH_SM((k_eb_walk<true>), dim3(N), dim3(128), wp_walk.lds, dj, wp_walk.vcw, geo_walk_pf());
    else LAUNCH_SM((k_eb_walk<false>), dim3(N), dim3(128), wp_walk.lds, dj, wp_walk.vcw, geo_walk_pf());
    LAUNCH(k_face_time, dim3(bf, N), dim3(UVOL_BLOCK), dj);
  }
  // valence replay + context scatter depend only on the walk: run them on the auxiliary stream, beside
  // renumber / seams / fans / DFS traversal on the main stream; joined again before the entropy stage.
  // The parallel kernels of the replay's preparation run on the MAIN stream, before the fork: beside the renumber / seams group they
  // took 25 + 39 ms of the auxiliary stream's time (its workgroups wait behind the main stream's 5 M-workgroup grids), which made
  // the auxiliary chain (125 ms) longer than the group it hides behind (70 ms) - the join below waited ~55 ms per batch.
  {
    LAUNCH(k_eb_event_flags, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_scan_sums, dim3(1, N), dim3(UVOL_BLOCK), dj, (int)SCAN_EVENTS);
    LAUNCH(k_eb_event_compact, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_valence_init, dim3(bc, N), dim3(UVOL_BLOCK), dj);
  }
  UVOL_HIP_CHECK(ctx, hipEventRecord(L.ev_walk, ctx->stream)); G->walk_last = L.ev_walk;
  UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(L.aux, L.ev_walk, 0));
  {
    { uvol_ctx::Scope sc(ctx, "geo.k4_eb_valence", 0, L.aux); LAUNCH_ON(L.aux, k_eb_valence, dim3(N), dim3(64), dj); }
    LAUNCH_ON(L.aux, k_eb_ctx, dim3(N), dim3(64), dj);
  }
  UVOL_HIP_CHECK(ctx, hipEventRecord(L.ev_val, L.aux));
  {
    uvol_ctx::Scope sc(ctx, "geo.k4b_seams", 0);
    LAUNCH(k_seams, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_sb_count, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_scan_sums, dim3(1, N), dim3(UVOL_BLOCK), dj, (int)SCAN_ELIG);
    LAUNCH(k_sb_write, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_aseg_a, dim3(bci, N, 2), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_aseg_b, dim3(bci, N, 2), dim3(UVOL_BLOCK), dj);
  }
  // The auxiliary stream (events / valence replay / context scatter, ~90 ms per 2160 frames) is joined HERE, not before the
  // entropy stage: it then overlaps the renumber / seams group only (about as long), but everything it reads (old-order
  // opposite corners and vertices, the symbol sequence, the valence scratch: 11.6 MB per frame) is dead before the three record
  // tables of the attribute traversals are written and shares their addresses - the workspace peak drops from 63 to 52 MB.
  if (!late_join) UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, L.ev_val, 0));
  {
    if (base_shared) LAUNCH(k_pack_tabs, dim3(bf, N, 2), dim3(UVOL_BLOCK), dj, fmtT, 2); else LAUNCH(k_pack_tabs, dim3(bf, N, 3), dim3(UVOL_BLOCK), dj, fmtT, 1);
    uvol_ctx::Scope sc(ctx, "geo.k5_traverse", (uint64_t)n * 32 * max_nfi * 3);
    // params.traverse_vbits_l2 (or UVOL_TRAVERSE_VGLOBAL=1): LDS traversers keep only the face bitmap in LDS (25 KB -> 6 per
    // CU instead of 3), the vertex bitmap lives in L2; each walker is ~30 % slower, twice as many are resident
    // unrelated meshes: FOUR traversers per wave in the wave form (3207 / 3774 / 4036 / 4060 / 3927 / 3780 frames/s geometry alone on two lanes with the lane
    // form at 1 and the wave form at 1 / 2 / 4 / 8 / 16 per wave, 2560 distinct frames: profiles/r05_walker_forms.json)
    // round 6: fewer per wave while the traversers of everything on the chip are few enough to have a SIMD (almost) to themselves - 1 / 2 / 4 per wave up to
    // 2048 / 4096 / more traversers: a 600-frame job 135 / 141 / 149 ms (1514 / 1490 / 1442 frames/s), a 300-frame job 122 / 125 / 127 ms (profiles/r06_small_job_forms.json)
    // (blocking calls only: enqueued calls run beside their predecessors' groups - a stream of 300-frame jobs lost 4 % with one per wave)
    if (wp_trav.simt_w > 1 && !lockstep && geo_simt_env() == 0) wp_trav.simt_w = fmtT == 2 ? ((G->blocking_call && 3u * NC <= 2048u) ? 1 : ((G->blocking_call && 3u * NC <= 4096u) ? 2 : 4)) : 1;
    if (w_trav_env && wp_trav.simt_w) wp_trav.simt_w = w_trav_env;
    launch_traversals(ctx, dj, n, wp_trav, fmtT, base_shared ? 1 : 0);
  }
  { uvol_ctx::Scope sc(ctx, "geo.k5b_v2d", 0); LAUNCH(k_v2d, dim3(be, N, 3), dim3(UVOL_BLOCK), dj, fmtT); }      // (own scope: geo.k5_traverse is exactly the traversal kernel, as rocprof lists it)
  {
    uvol_ctx::Scope sc(ctx, "geo.k6_predict", 0);
    LAUNCH(k_stream_setup, dim3(N), dim3(64), dj);
    LAUNCH(k_pred_pos, dim3(be, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_pred_uv, dim3(be, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_scan_blocks, dim3(be, N), dim3(UVOL_BLOCK), dj, (int)SCAN_ORI);
    LAUNCH(k_scan_sums, dim3(1, N), dim3(UVOL_BLOCK), dj, (int)SCAN_ORI);
    LAUNCH(k_ori_compact, dim3(be, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_ori_bits, dim3(be, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_face_normals, dim3(bf, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_pred_nrm, dim3(be, N), dim3(UVOL_BLOCK), dj);
  }
  if (late_join) UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, L.ev_val, 0));
  {
    uvol_ctx::Scope sc0(ctx, "geo.k7_hist_tables", 0);
    LAUNCH(k_hist, dim3(uvol_blocks((size_t)9 * max_nfi, 16 * UVOL_BLOCK), GEO_NSTREAM, N), dim3(UVOL_BLOCK), dj);
    LAUNCH(k_rans_tables, dim3(GEO_NSTREAM, N), dim3(64), dj);
  }
  {
    uvol_ctx::Scope sc(ctx, "geo.k7_entropy_encode", 0);
    // Wave-per-stream coder (state recurrence on the scalar unit, ~90 ns per symbol) while all 14 x N one-wave workgroups are resident
    // at once (8 KiB of LDS each: 20 per CU); beyond that the lane-per-stream coder, which is slower per stream (~175 ns per symbol)
    // but runs every stream of the batch concurrently (300 frames: 28 vs 72 ms; 2160 frames: 53 ms with lanes).
    // UVOL_ENTROPY_WAVE=1 / 0 (tests / diagnostic) forces one form.
    static const int ent_env = [] { const char *e = getenv("UVOL_ENTROPY_WAVE"); return !e ? -1 : (*e == '1' ? 1 : 0); }();
    static const int ent_w_env = [] { const char *e = getenv("UVOL_ENTROPY_W"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();   // lanes per wave of the lane form (implies it)
    const bool ent_wave = ent_env >= 0 ? ent_env == 1 : (ent_w_env == 0 && (size_t)(GEO_NSTREAM + GEO_NRABS) * NC <= (size_t)20 * G->num_cu);
    if (ent_wave) LAUNCH(k_entropy_encode, dim3(N, GEO_NSTREAM + GEO_NRABS), dim3(64), dj, uvol_debug() ? 1 : 0);      // frame index fastest: the long streams of the frames spread over the four SIMDs of a CU (geom_decode.hip: k_gdec_rans)
    else {
      // lanes per wave: five streams of a frame are long (three attribute symbol streams, two seam-bit streams; ~300 k steps) and a
      // wave runs as long as its longest lane, so the launch should put at most ONE long wave on a SIMD (1024 of them): waves that
      // share a SIMD share its issue slots (2160 frames: 172 / 105 / 60 / 64 / 55 / 52 ms with 1 / 2 / 4 / 8 / 16 / 32 lanes per wave)
      unsigned W = 4; while (W < 64 && 5u * NC > 512u * W) W *= 2;
      if (ent_w_env) W = (unsigned)ent_w_env;
      LAUNCH(k_rans_recip, dim3(uvol_blocks(((size_t)2 << std::max(prm.q_position_attr, std::max(prm.q_texture_attr, prm.q_normal_attr))) + 8), GEO_NSTREAM, N), dim3(UVOL_BLOCK), dj);
      LAUNCH_SM(k_entropy_simt, dim3((N + W - 1) / W, GEO_NSTREAM + GEO_NRABS), dim3(64), (size_t)W * SB_STRIDE * 4, dj, n, (int)W);
    }
  }
  }   // !seq
  {
    uvol_ctx::Scope sc(ctx, "geo.k8_layout_gather", 0);
    if (seq) LAUNCH(k_sq_layout, dim3(N), dim3(64), dj); else LAUNCH(k_layout, dim3(N), dim3(64), dj);
    LAUNCH(k_out_offsets, dim3(1), dim3(64), dj, n);
    LAUNCH(k_gather, dim3(64, GEO_MAXPIECES, N), dim3(UVOL_BLOCK), dj);
  }
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  if (pre_up) { const int rcr = uvol_uplink_release(ctx, L.up.slot, ctx->stream); if (rcr != UVOL_OK) return rcr; }      // (the auxiliary stream has been joined: nothing of this group reads the slot later)
  // (the job records are read back by geo_complete_impl: a device-to-host copy into pageable memory does not return before the stream
  // has reached it, i.e. before every kernel of this group is done - here it serialised the groups of a call)
  L.t_enq = ms_since(t_enter); L.t_enter = t_enter;
  return UVOL_OK;
}

// Second half of a group: waits for its stream, copies the packed bitstreams out (one device-to-host copy into pinned staging, then
// plain memcpy into the caller's buffers), fills out_lens / status, re-encodes the frames the compact workspace could not hold.
static int geo_complete_impl(uvol_ctx *ctx, GeoLane &L) {
  static const bool timing = [] { const char *e = getenv("UVOL_TIMING"); return e && *e == '1'; }();       // diagnostic: host-side phases of a batch on stderr
  auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  const int n = L.n; const bool full = L.full, on_device = L.on_device;
  uint8_t *const *outs = L.outp.data(); size_t *out_lens = L.out_lens; int *status = L.status;
  const auto t_enter = L.t_enter; const double t_prep = L.t_prep, t_enq = L.t_enq;
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  UVOL_HIP_CHECK(ctx, hipMemcpy(L.hjobs.data(), L.jobs.p, sizeof(GeoJob) * (size_t)n, hipMemcpyDeviceToHost));
  const double t_gpu = ms_since(t_enter);
  int worst = UVOL_OK;
  // one device-to-host copy of the packed bitstreams into pinned staging, then plain memcpy into the caller's buffers
  size_t packed = 0;
  for (int i = 0; i < n; i++) { const GeoJob &J = L.hjobs[i]; if (J.status == 0) packed = std::max<size_t>(packed, (size_t)J.out_pack_off + J.out_len); }
  if (L.ext_out) packed = 0;                               // GPU-resident form: the bitstreams stay where k_gather put them
  if (packed > L.pinned_cap) {
    if (L.pinned) (void)hipHostFree(L.pinned);
    L.pinned = nullptr; L.pinned_cap = 0;
    const size_t want = packed + packed / 4 + (1u << 20);
    UVOL_HIP_CHECK(ctx, hipHostMalloc((void **)&L.pinned, want, hipHostMallocDefault));
    L.pinned_cap = want;
    // (the ring's lanes that have their device buffers but no staging yet get it now, for the same reason as in geo_encode_batch_begin)
    for (GeoLane *o : ctx->geo->lanes) if (o != &L && !o->pinned && o->slab.p && !o->ext_out) { if (hipHostMalloc((void **)&o->pinned, want, hipHostMallocDefault) == hipSuccess) o->pinned_cap = want; else { (void)hipGetLastError(); o->pinned = nullptr; } }
  }
  if (packed) {
    UVOL_HIP_CHECK(ctx, hipMemcpyAsync(L.pinned, L.outs.p, packed, hipMemcpyDeviceToHost, ctx->stream));
    UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  const double t_d2h = ms_since(t_enter);
  std::vector<int> retry;
  // staging -> the caller's buffers: a few host threads for large batches (2160 frames x 250 KB took ~60 ms of one core per batch)
  { const int nt = packed > ((size_t)32 << 20) ? 8 : 1;
    auto copy_range = [&](int a, int b) { for (int i = a; i < b; i++) { const GeoJob &J = L.hjobs[i]; if (J.status == 0 && !L.ext_out) memcpy(outs[i], L.pinned + J.out_pack_off, J.out_len); } };
    if (nt == 1) copy_range(0, n);
    else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(copy_range, (int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt)); for (auto &x : th) x.join(); } }
  for (int i = 0; i < n; i++) {
    const GeoJob &J = L.hjobs[i];
    int st = J.status == 0 ? UVOL_OK : (J.status == UVOL_E_NOSPACE ? UVOL_E_NOSPACE : UVOL_E_ENCODE);
    out_lens[i] = J.out_len;
    if (L.ext_offs) L.ext_offs[i] = (size_t)J.out_pack_off;
    if (st == UVOL_OK) { }
    else if (!full && (J.status == GEO_E_WS_OVERFLOW || J.status == GEO_E_SLAB_FULL || J.status == GEO_E_DD_OVERFLOW)) { retry.push_back(i); st = UVOL_OK; }
    else { ctx->set_error("mesh %d: encode failed (device status %d)", i, J.status); worst = st; }
    if (status) status[i] = st;
  }
  // frames the compact workspace (or the packed output area) could not hold: once more, alone, with worst-case sizes
  if (!retry.empty()) {
    // the lane's own record of the group is replaced by the one-frame retries: keep what is still needed
    const std::vector<uvol_mesh> rm(L.meshes); const std::vector<uint8_t *> ro(L.outp); const std::vector<size_t> rcap(L.caps);
    // GPU-resident form: a retried frame goes behind the frames already packed into the caller's buffer
    uint8_t *const ext0 = L.ext_out; const size_t ext_cap0 = L.ext_cap; size_t *const offs0 = L.ext_offs; size_t tail = 0;
    if (ext0) for (int i = 0; i < n; i++) { const GeoJob &J = L.hjobs[i]; if (J.status == 0) tail = std::max<size_t>(tail, ((size_t)J.out_pack_off + J.out_len + 255) & ~(size_t)255); }
    for (int i : retry) {
      if (timing) fprintf(stderr, "[uvol-timing] mesh %d: re-encoding with worst-case workspace\n", i);
      int st1 = UVOL_OK;
      if (ext0) {
        if (tail >= ext_cap0) { if (status) status[i] = UVOL_E_NOSPACE; worst = UVOL_E_NOSPACE; out_lens[i] = 0; continue; }
        L.ext_out = ext0 + tail; L.ext_cap = ext_cap0 - tail; L.ext_offs = nullptr; L.producer = nullptr;
      }
      size_t cap1 = ext0 ? std::min<size_t>(ext_cap0 - tail, 0xffffffffu) : rcap[i];
      int rc1 = geo_submit(ctx, L, &rm[i], 1, 1, on_device, &ro[i], &cap1, out_lens + i, &st1, true);
      if (rc1 == UVOL_OK) rc1 = geo_complete(ctx, L);
      if (ext0) { L.ext_out = ext0; L.ext_cap = ext_cap0; L.ext_offs = offs0; if (offs0) offs0[i] = tail; if (rc1 == UVOL_OK && st1 == UVOL_OK) tail = (tail + out_lens[i] + 255) & ~(size_t)255; }
      if (rc1 != UVOL_OK) return rc1;
      if (status) status[i] = st1;
      if (st1 != UVOL_OK) worst = st1;
    }
  }
  if (timing) fprintf(stderr, "[uvol-timing] geo group n=%d sizeof(GeoJob)=%zu: host prepared %.1f ms, enqueued %.1f, gpu done %.1f, packed d2h %.1f, copied out %.1f (enter at %.1f)\n", n, sizeof(GeoJob), t_prep, t_enq, t_gpu, t_d2h, ms_since(t_enter),
                      std::chrono::duration<double, std::milli>(t_enter.time_since_epoch()).count());
  return status ? UVOL_OK : worst;
}

// the lane's streams stand in for the context's while one of its groups is submitted / completed (LAUNCH, Scope, uvol_ensure use ctx->stream)
static int geo_submit(uvol_ctx *ctx, GeoLane &L, const uvol_mesh *meshes, int n, int n_conc, bool on_device,
                      uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status, bool full, GeoUp *pre) {
  if (pre) L.up = std::move(*pre); else { L.up.slot = nullptr; L.up.off.clear(); }
  L.meshes.assign(meshes, meshes + n); L.outp.assign(outs, outs + n); L.caps.assign(caps, caps + n);
  L.out_lens = out_lens; L.status = status; L.n = n; L.n_conc = n_conc; L.on_device = on_device; L.full = full;
  hipStream_t saved = ctx->stream; ctx->stream = L.stream;
  const int rc = geo_submit_impl(ctx, L, L.meshes.data(), n, n_conc, on_device, L.caps.data(), full);
  ctx->stream = saved;
  if (rc != UVOL_OK && L.up.slot) { (void)hipStreamSynchronize(L.stream); (void)hipStreamSynchronize(L.aux); L.up.slot = nullptr; }      // (kernels already enqueued may read the slot: its release was never recorded)
  L.busy = rc == UVOL_OK;
  return rc;
}
static int geo_complete(uvol_ctx *ctx, GeoLane &L) {
  if (!L.busy) return UVOL_OK;
  L.busy = false;
  hipStream_t saved = ctx->stream; ctx->stream = L.stream;
  const int rc = geo_complete_impl(ctx, L);
  ctx->stream = saved;
  return rc;
}

// completes every group still in flight (enqueued calls complete lazily, so that the next call's front end overlaps this call's walkers);
// returns the first error among them and among the groups completed earlier on behalf of later calls
int geo_flush(uvol_ctx *ctx) {
  GeoState *G = ctx->geo; if (!G) return UVOL_OK;
  int rc = G->deferred_rc; G->deferred_rc = UVOL_OK;
  const int nl = (int)G->lanes.size();
  for (int k = 0; k < nl; k++) { GeoLane *L = G->lanes[(G->next_lane + k) % nl]; const int r = geo_complete(ctx, *L); if (rc == UVOL_OK) rc = r; }
  ctx->resolve_profile();
  return rc;
}

// lanes of a context: UVOL_GEO_LANES (1 = every call one group on the context's stream, as before round 4).  Measured in round 4 on 2560
// distinct frames per call, enqueued calls: 1 / 2 / 3 / 4 lanes = 3176 / 3433 / 3401 / 3420 frames/s geometry alone, 2607 / 2682 / 2421 / 2517
// beside the texture context - two groups overlap their front ends and walkers, more only add interference - while a blocking call cut
// into four groups was SLOWER than one group (2454 against 3176: nothing runs beside the last group's walkers, and every group pays
// its own read-back).
// Round 5: the ring is LONGER than the groups of one call - SIX lanes, an enqueued call of device inputs cut into FOUR groups (UVOL_GEO_GROUPS) - so
// the lanes past the fourth take the first groups of the NEXT enqueued call: 1.5 calls' worth of frames are on the chip while the caller keeps ONE
// call's inputs resident, and six groups at different points of their chains mix streaming and walker phases more evenly than three.  2560 distinct
// frames per call, frames/s geometry alone / beside the texture context: 2 lanes x 2 groups 4035 / 3118 - 3161, 3 x 2 4964 / 3263 - 3397, 6 x 4
// 5307 / 3525 - 3650, 7 x 4 - / 3743 (244 GB of HBM in use), 8 x 5 5435 / 3583, 12 x 8 - / 3194 (profiles/r05_frames_in_flight.json,
// r05_ring_shapes.json).  A lane that cannot get its workspace leaves the ring (GeoState::lanes_cap).
static inline int geo_lanes_wanted() { static const int v = [] { const char *e = getenv("UVOL_GEO_LANES"); const int k = e ? atoi(e) : 6; return k < 1 ? 1 : (k > 16 ? 16 : k); }(); return v; }
// frames per group at least (UVOL_GEO_MIN_GROUP, tests: small values spread small calls over the lanes): below 2 x this a call stays one
// group - its walkers are the whole critical path anyway
static inline int geo_min_group() { static const int v = [] { const char *e = getenv("UVOL_GEO_MIN_GROUP"); const int k = e ? atoi(e) : 160; return k < 1 ? 1 : k; }(); return v; }

// Enqueue n frames: the call is cut into up to `lanes` contiguous groups, each submitted on the next lane of the ring.  A lane that still
// holds a group of an EARLIER call is completed first (its error, if any, is kept for geo_flush).
// split: cut the call into groups (enqueued calls, whose successor overlaps their tail; blocking calls with HOST inputs, whose groups
// upload while the groups before them encode); a blocking call on device inputs stays one group.
int geo_encode_batch_begin(uvol_ctx *ctx, const uvol_mesh *meshes, int n, bool on_device,
                           uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status, bool split) {
  GeoState *G = ctx->geo;
  if (n <= 0) return UVOL_OK;
  // host inputs: four groups at least (a group uploads while the groups before it encode; the first group's upload is the only one nothing hides)
  const int want = std::max(1, std::min(G->lanes_cap, on_device ? geo_lanes_wanted() : std::max(geo_lanes_wanted(), 4)));
  static const int split_env = [] { const char *e = getenv("UVOL_GEO_SPLIT"); return e ? atoi(e) : -1; }();      // tests / diagnostic: 1 / 0 force / forbid the split
  if (split_env >= 0) split = split_env != 0;
  // groups per call <= lanes: with device inputs a call is cut into UVOL_GEO_GROUPS (default 4) groups while the ring has `want` lanes, so
  // that consecutive enqueued calls hold want / groups calls' worth of frames on the chip (the walkers' chain is flat in the frame count:
  // throughput follows the frames in flight, profiles/r05_frames_in_flight.json) without the caller keeping more inputs resident
  static const int groups_env = [] { const char *e = getenv("UVOL_GEO_GROUPS"); const int k = e ? atoi(e) : 4; return k < 1 ? 1 : k; }();
  // Host inputs that ALL lie in uvol_host_alloc memory travel through the uplink (below): their upload does not occupy a lane, so such a call is
  // cut like a call on device inputs.  (Round 5 cut every host call into as many groups as the ring has lanes: consecutive enqueued calls then
  // met lane by lane - group g of call c + 1 waited for group g of call c - and the six groups of a call ran their walkers, then their
  // traversals, side by side, bunched: 2333 frames/s geometry alone against 5307 on device inputs, profiles/r06_uplink_forms.json.)
  bool up_all = !on_device && uvol_uplink_enabled() && split && n >= 1;
  for (int i = 0; i < n && up_all; i++) {
    const uvol_mesh &m = meshes[i];
    if (!m.pos || !m.idx_pos || m.n_pos == 0 || m.n_faces == 0) { up_all = false; break; }                    // (geo_submit reports the invalid frame)
    up_all = uvol_host_pinned(m.pos, (size_t)m.n_pos * 12) && uvol_host_pinned(m.idx_pos, (size_t)m.n_faces * 12);
    if (up_all && m.uv && m.idx_uv && m.n_uv) up_all = uvol_host_pinned(m.uv, (size_t)m.n_uv * 8) && uvol_host_pinned(m.idx_uv, (size_t)m.n_faces * 12);
    if (up_all && m.nrm && m.idx_nrm && m.n_nrm) up_all = uvol_host_pinned(m.nrm, (size_t)m.n_nrm * 12) && uvol_host_pinned(m.idx_nrm, (size_t)m.n_faces * 12);
  }
  const int gmax = (on_device || up_all) ? std::min(want, groups_env) : want;
  const int groups = split ? std::max(1, std::min(gmax, n / geo_min_group())) : 1;
  // Inputs in uvol_host_alloc memory (SURVEY 8(d)'s boundary): the uploads of ALL groups of the call are queued on the context's copy
  // stream now, one uplink slot per group, before the first group's kernels are enqueued (uvol_common.hpp "Uplink").  The ring has as
  // many slots as the lane ring (>= groups), so the slot a group takes was last used by a group of an EARLIER call, whose kernels - and
  // with them the slot's release event - have been enqueued.  A frame's arrays are placed in the order of their host addresses, the
  // frames in call order: a caller that lays consecutive frames back to back in its arena gets one DMA per run of frames.
  std::vector<GeoUp> ups_pre;
  if (up_all) {
    const bool all = true;
    UvolUplink *U = all ? uvol_uplink(ctx, (size_t)std::max(want + uvol_uplink_ahead(), groups)) : nullptr;
    if (all && !U) return UVOL_E_HIP;
    if (U) {
      ups_pre.resize((size_t)groups);
      std::vector<UvolUpItem> items;
      for (int g = 0; g < groups; g++) {
        const int a = (int)((long long)n * g / groups), b = (int)((long long)n * (g + 1) / groups);
        GeoUp &P = ups_pre[(size_t)g]; P.off.assign((size_t)(b - a) * 6, 0);
        UvolUpPlacer pl; items.clear(); items.reserve((size_t)(b - a) * 6);
        for (int i = a; i < b; i++) {
          const uvol_mesh &m = meshes[i];
          const bool hu = m.uv && m.idx_uv && m.n_uv, hn = m.nrm && m.idx_nrm && m.n_nrm;
          struct Arr { const void *src; size_t bytes; int k; } arr[6] = {
            { m.pos, (size_t)m.n_pos * 12, 0 }, { hu ? m.uv : nullptr, hu ? (size_t)m.n_uv * 8 : 0, 1 }, { hn ? m.nrm : nullptr, hn ? (size_t)m.n_nrm * 12 : 0, 2 },
            { m.idx_pos, (size_t)m.n_faces * 12, 3 }, { hu ? m.idx_uv : nullptr, hu ? (size_t)m.n_faces * 12 : 0, 4 }, { hn ? m.idx_nrm : nullptr, hn ? (size_t)m.n_faces * 12 : 0, 5 } };
          std::sort(arr, arr + 6, [](const Arr &x, const Arr &y) { return (uintptr_t)x.src < (uintptr_t)y.src; });
          for (const Arr &A : arr) if (A.src && A.bytes) { const size_t d = pl.place(A.src, A.bytes); P.off[(size_t)(i - a) * 6 + A.k] = d; items.push_back(UvolUpItem{ d, A.src, A.bytes }); }
        }
        P.slot = uvol_uplink_fill(ctx, U, items, pl.total());
        if (!P.slot) return UVOL_E_HIP;
      }
    }
  }
  GeoLane *last = nullptr;
  for (int g = 0; g < groups; g++) {
    const int a = (int)((long long)n * g / groups), b = (int)((long long)n * (g + 1) / groups);
    // a blocking call on device inputs always runs on lane 0 (one workspace of its size per context, as before); the others take the ring
    GeoLane *L = geo_lane(ctx, split ? G->next_lane % want : 0);
    if (!L) { ctx->set_error("geometry lane: stream / event creation failed"); return UVOL_E_HIP; }
    if (split) G->next_lane = (G->next_lane + 1) % want;
    if (L->busy) { const int r = geo_complete(ctx, *L); if (r != UVOL_OK && G->deferred_rc == UVOL_OK) G->deferred_rc = r; }
    const int rc = geo_submit(ctx, *L, meshes + a, b - a, n, on_device, outs + a, caps + a, out_lens + a, status ? status + a : nullptr, false, ups_pre.empty() ? nullptr : &ups_pre[(size_t)g]);
    if (rc != UVOL_OK) {
      // the slots this call filled and nobody will read: their copies must not outlive the caller's arrays (the call has failed)
      if (!ups_pre.empty() && ctx->uplink) (void)hipStreamSynchronize(ctx->uplink->stream);
      return rc;
    }
    last = L;
  }
  // The lanes of the ring that have not held a group yet get their buffers NOW, sized like the group just submitted: a first allocation of tens of
  // GB takes the runtime from milliseconds to more than a second (profiles/r05_x_regimes.json: 1.1 - 1.6 s per lane in every other process on
  // this pool), and a ring longer than the groups of one call first reaches its last lane during the caller's SECOND call - here it happens
  // beside the kernels of the first.  A lane that cannot get them leaves the ring (lanes_cap).
  // (only the lanes the NEXT call of this shape will reach: a stream of small calls - one group each - allocates one lane ahead, not the whole ring at once)
  if (split && (on_device || up_all) && last && groups < want) {
    for (int j = 0; j < std::min(groups, want); j++) {
      const int k = (G->next_lane + j) % want;
      GeoLane *P = geo_lane(ctx, k);
      if (!P || P == last || P->busy || P->slab.p) continue;
      bool ok = true;
      for (auto pr : { std::make_pair(&P->slab, last->slab.cap), std::make_pair(&P->outs, last->outs.cap), std::make_pair(&P->jobs, last->jobs.cap) })
        if (ok && pr.second && !pr.first->p) { if (hipMalloc(&pr.first->p, pr.second) == hipSuccess) pr.first->cap = pr.second; else { (void)hipGetLastError(); pr.first->p = nullptr; ok = false; } }
      if (!ok) { for (uvol_devbuf *b : { &P->slab, &P->outs, &P->jobs }) if (b->p) { (void)hipFree(b->p); b->p = nullptr; b->cap = 0; } int have = 0; for (GeoLane *o : G->lanes) have += o->slab.p ? 1 : 0; G->lanes_cap = std::max(1, std::min(G->lanes_cap, have)); break; }
    }
  }
  return UVOL_OK;
}
// GPU-resident form: one group on lane 0 (the packed output area is one caller buffer), ordered after `producer`
int geo_encode_batch_dev_out(uvol_ctx *ctx, const uvol_mesh *meshes, int n, hipStream_t producer, uint8_t *dev_out, size_t dev_cap, size_t *out_offs, size_t *out_lens, int *status) {
  if (n <= 0) return UVOL_OK;
  int rf = geo_flush(ctx);                                 // nothing else of this context in flight
  GeoLane *L = geo_lane(ctx, 0);
  if (!L) { ctx->set_error("geometry lane: stream / event creation failed"); return UVOL_E_HIP; }
  std::vector<uint8_t *> outs((size_t)n, nullptr); std::vector<size_t> caps((size_t)n, std::min<size_t>(dev_cap, 0xffffffffu));
  L->ext_out = dev_out; L->ext_cap = dev_cap; L->ext_offs = out_offs; L->producer = producer;
  int rc = geo_submit(ctx, *L, meshes, n, n, true, outs.data(), caps.data(), out_lens, status, false);
  if (rc == UVOL_OK) rc = geo_complete(ctx, *L);
  L->ext_out = nullptr; L->ext_cap = 0; L->ext_offs = nullptr; L->producer = nullptr;
  ctx->resolve_profile();
  return rf != UVOL_OK ? rf : rc;
}
// blocking form: begin + flush
int geo_encode_batch(uvol_ctx *ctx, const uvol_mesh *meshes, int n, bool on_device,
                     uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status) {
  ctx->geo->blocking_call = true;
  const int rc = geo_encode_batch_begin(ctx, meshes, n, on_device, outs, caps, out_lens, status, !on_device);
  ctx->geo->blocking_call = false;
  const int rf = geo_flush(ctx);
  return rc != UVOL_OK ? rc : rf;
}
