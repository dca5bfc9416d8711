// uvol_common.hpp — shared host-side plumbing of libuvolcodec (ctx, error handling, device arena,
// per-kernel-group hipEvent profiling).  gfx950 only; no CPU fallback anywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>
#include "../../include/uvol_codec.h"

#define UVOL_BLOCK 256

// wave-uniform lane read (v_readlane on the GPU, a shuffle in the shim)
#ifdef HIPEMU
#define UVOL_READLANE(v, l) ((uint32_t)__shfl((uint32_t)(v), (int)(l)))
#else
#define UVOL_READLANE(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(l)))
#endif
// the wave-uniform `val` into lane `l` (uniform) of a per-lane register, the other lanes keep `old` (v_cmp + v_cndmask; this
// toolchain has no v_writelane builtin)
#define UVOL_WRITELANE(val, l, old) ((int)(threadIdx.x & 63) == (int)(l) ? (uint32_t)(val) : (uint32_t)(old))

// intra-wave ordering point between lane 0's stores and the other lanes' loads: lock-step on the GPU (plus a
// compiler barrier); a real rendezvous in the shim, where lanes are fibers that run ahead of each other
#ifdef HIPEMU
#define UVOL_WAVE_SYNC() do { (void)__shfl(0, 0); } while (0)
#else
#define UVOL_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

// ordering point for data only ONE wave reads and writes (the valence replay's per-frame arrays): its own memory operations
// complete before it goes on (s_waitcnt).  __threadfence() here meant an L2 write-back + L1 invalidate per 64 symbols and wave
// (~3.5 us each, MI355X_MICROARCH.md), which also slowed every kernel running beside it: k_seams took 270 instead of 30 ms.
#ifdef HIPEMU
#define UVOL_WAVE_FENCE() std::atomic_thread_fence(std::memory_order_seq_cst)
#else
#define UVOL_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#endif

// the one-wave serial walkers are latency-bound: give them issue priority over co-resident throughput kernels
#ifdef HIPEMU
#define UVOL_SERIAL_PRIO() do { } while (0)
#else
#define UVOL_SERIAL_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

// Address-space-qualified pointers for the serial walkers.  A pointer read out of a job struct is "generic", so the
// compiler has to use flat_* instructions, which count in BOTH vmcnt and lgkmcnt: every LDS bitmap test then also
// waits for the global stores issued just before it (a full memory round trip per face).  Casting to the global /
// LDS address spaces yields global_* / ds_* instructions with independent, exactly counted waits.
#ifdef HIPEMU
#define UVOL_G(T) T *
#define UVOL_L(T) T *
#define UVOL_TO_G(T, p) (p)
#define UVOL_TO_L(T, p) (p)
#define UVOL_OR_NORET(p, v) ((void)(*(p) |= (v)))
#else
#define UVOL_G(T) __attribute__((address_space(1))) T *
#define UVOL_L(T) __attribute__((address_space(3))) T *
#define UVOL_TO_G(T, p) ((__attribute__((address_space(1))) T *)(p))
#define UVOL_TO_L(T, p) ((__attribute__((address_space(3))) T *)(p))
#define UVOL_OR_NORET(p, v) ((void)__hip_atomic_fetch_or((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#endif

// device-scope relaxed atomics on global words (L2-coherent: loads bypass the per-CU L1) for data that lanes of one wave
// update with atomic adds and read back later; plain operations in the shim (one OS thread runs a whole workgroup)
#ifdef HIPEMU
#define UVOL_ALOAD(p) (*(p))
#define UVOL_ASTORE(p, v) ((void)(*(p) = (v)))
#define UVOL_AADD(p, v) ((void)(*(p) += (v)))
#else
#define UVOL_ALOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define UVOL_ASTORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define UVOL_AADD(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#endif

// Prefetching in a one-lane walk.  The walk is provably wave-uniform, so hipcc moves every loaded value to SGPRs with
// v_readfirstlane right after the load — and the s_waitcnt that needs turns a prefetch into a stall.  Adding
// UVOL_LANE_ZERO() (the lane id of lane 0, which the compiler cannot see through) to the prefetch ADDRESS keeps the
// result in VGPRs; UVOL_READFIRST() moves it to scalars only where the value is finally consumed, so the load overlaps
// the step's bookkeeping (tools/latbench variants 11 / 12: 894 -> 581 ns per step).
#ifdef HIPEMU
#define UVOL_LANE_ZERO() 0
#define UVOL_READFIRST(v) (v)
#define UVOL_BCAST0(v) (__shfl((v), 0))
#define UVOL_OPAQUE(v) ((void)0)
#else
#define UVOL_LANE_ZERO() ((int)__builtin_amdgcn_mbcnt_lo(~0u, 0u))
#define UVOL_READFIRST(v) (__builtin_amdgcn_readfirstlane((int)(v)))
// lane 0's value in every lane of a FULLY active wave (the cooperative walkers): readfirstlane on the GPU; in the shim a real
// exchange, which also is the rendezvous that keeps its free-running lanes from acting on LDS words a faster lane already changed
#define UVOL_BCAST0(v) (__builtin_amdgcn_readfirstlane((int)(v)))
// empty asm the optimiser cannot look through: stops it re-associating across `v` (emits no instruction)
#define UVOL_OPAQUE(v) asm volatile("" : "+v"(v))
#endif

// value of the lane one below within a row of 16 lanes (DPP row_shr:1: a register move, no LDS round trip; lanes 0 / 16 / 32 / 48 get 0
// on the GPU and an unrelated value in the shim - callers do not use it there)
#ifdef HIPEMU
#define UVOL_ROW_SHR1(v) ((uint32_t)__shfl_up((int)(v), 1))
#else
#define UVOL_ROW_SHR1(v) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x111, 0xf, 0xf, false))
#endif

// occupancy target of a kernel (waves per SIMD): an attribute hipcc understands, nothing in the tests/hipemu build
#ifdef HIPEMU
#define UVOL_WAVES_PER_EU(n)
#else
#define UVOL_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif

// dynamic LDS: `extern __shared__` on the GPU, the shim's per-workgroup buffer in the tests/hipemu build
#ifdef HIPEMU
#define UVOL_DYN_SMEM(T, name) T *name = reinterpret_cast<T *>(hipemu_dyn_smem)
#else
#define UVOL_DYN_SMEM(T, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw_[]; T *name = reinterpret_cast<T *>(name##_raw_)
#endif

#define UVOL_HIP_CHECK(ctx, expr)                                                                  \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      (ctx)->set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return UVOL_E_HIP;                                                                           \
    }                                                                                              \
  } while (0)

struct uvol_prof_entry {
  std::string name;
  uint64_t launches = 0;
  double total_ms = 0;
  uint64_t algo_bytes = 0;
};
struct uvol_prof_pending { int idx; hipEvent_t a, b; };

// A growable device allocation that is re-used across calls (no hipMalloc in the steady state).
struct uvol_devbuf {
  void *p = nullptr;
  size_t cap = 0;
};

struct GeoState;   // geometry pipeline state (geom_encode.hip)
struct TexState;   // texture pipeline state (tex_encode.hip)
struct TexDecState;  // texture decode state (tex_decode.hip)
struct GeoDecState;  // geometry decode state (geom_decode.hip)
struct UastcState;   // UASTC texture mode (tex_uastc.hip)
struct ObjState;     // OBJ text ingest on the device (obj_ingest.hip)
struct PngState;     // PNG scanlines un-filtered on the device (png_ingest.hip)

struct uvol_ctx {
  int device = 0;
  uvol_params prm{};
  hipStream_t stream = nullptr;
  char err[512] = {0};
  bool profiling = false;
  std::vector<uvol_prof_entry> prof;
  std::vector<uvol_prof_pending> pending;
  std::vector<hipEvent_t> event_pool;
  GeoState *geo = nullptr;
  TexState *tex = nullptr;
  TexDecState *texdec = nullptr;
  GeoDecState *geodec = nullptr;
  UastcState *uastc = nullptr;
  ObjState *obj = nullptr;
  PngState *png = nullptr;
  // enqueue form of the ABI (uvol_*_async + uvol_sync): calls run in order on this context's worker thread
  struct AsyncQ {
    std::mutex m; std::condition_variable cv_work, cv_idle; std::deque<std::function<int()>> q; std::thread th; bool busy = false, stop = false; int first_err = 0; char err[512] = {0};
  } *async = nullptr;
  uint8_t *dn_pin[2] = { nullptr, nullptr }; hipEvent_t dn_ev[2] = { nullptr, nullptr };      // staged downloads (uvol_download_staged)
  hipEvent_t pin_ev[2] = { nullptr, nullptr };          // uploads from uvol_host_alloc memory: one event per queued run of copies
  struct UvolUplink *uplink = nullptr;                   // upload ring for inputs in uvol_host_alloc memory (uvol_uplink_*, below); freed by uvol_uplink_destroy
  uint8_t *up_pin[2] = { nullptr, nullptr }; size_t up_cap = 0; hipEvent_t up_ev[2] = { nullptr, nullptr }; bool up_rec[2] = { false, false };   // staged uploads (uvol_upload_staged); up_rec: a DMA out of that buffer may still be in flight

  void set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(err, sizeof(err), fmt, ap); va_end(ap);
  }
  int prof_index(const char *name) {
    for (size_t i = 0; i < prof.size(); i++) if (prof[i].name == name) return (int)i;
    prof.push_back({name, 0, 0.0, 0});
    return (int)prof.size() - 1;
  }
  hipEvent_t get_event() {
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; return e;
  }
  // bracket a kernel group; resolved lazily at the next sync
  struct Scope {
    uvol_ctx *c; int idx; hipEvent_t a = nullptr, b = nullptr; hipStream_t st;
    Scope(uvol_ctx *ctx, const char *name, uint64_t algo_bytes, hipStream_t on = nullptr) : c(ctx), idx(-1), st(on ? on : ctx->stream) {
      if (!c->profiling) return;
      idx = c->prof_index(name); c->prof[idx].launches++; c->prof[idx].algo_bytes += algo_bytes;
      a = c->get_event(); b = c->get_event();
      if (a) (void)hipEventRecord(a, st);
    }
    ~Scope() { if (idx >= 0 && a && b) { (void)hipEventRecord(b, st); c->pending.push_back({idx, a, b}); } }
  };
  void resolve_profile() {
    for (auto &p : pending) {
      float ms = 0;
      if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) prof[p.idx].total_ms += ms;
      event_pool.push_back(p.a); event_pool.push_back(p.b);
    }
    pending.clear();
  }
};

static inline int uvol_ensure(uvol_ctx *ctx, uvol_devbuf &b, size_t bytes) {
  if (bytes <= b.cap) return UVOL_OK;
  if (b.p) { UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); UVOL_HIP_CHECK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
  size_t want = bytes + std::min<size_t>(bytes / 8, (size_t)256 << 20) + 4096;      // slack against re-allocation for slightly larger batches; bounded: an eighth of a 100 GB workspace is frames that could be in flight
  UVOL_HIP_CHECK(ctx, hipMalloc(&b.p, want));
  b.cap = want;
  return UVOL_OK;
}

static inline bool uvol_debug() { static int d = -1; if (d < 0) { const char *e = getenv("UVOL_DEBUG"); d = (e && *e && *e != '0') ? 1 : 0; } return d == 1; }
static inline unsigned uvol_blocks(size_t n, unsigned bs = UVOL_BLOCK) { return (unsigned)((n + bs - 1) / bs); }

// pipeline entry points implemented in the .hip translation units
hipError_t uvol_make_stream(uvol_ctx *ctx, hipStream_t *out);
int geo_create(uvol_ctx *ctx);
void geo_destroy(uvol_ctx *ctx);
int geo_encode_batch(uvol_ctx *ctx, const uvol_mesh *meshes, int n, bool inputs_on_device,
                     uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status);
// enqueue form: the groups of the call are submitted to the context's lanes and completed by geo_flush (or when a later call needs the lane)
int geo_encode_batch_begin(uvol_ctx *ctx, const uvol_mesh *meshes, int n, bool inputs_on_device,
                           uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status, bool split);
int geo_flush(uvol_ctx *ctx);
int geo_trim(uvol_ctx *ctx);
// GPU-resident form: one group on lane 0, ordered after `producer`, bitstreams packed into the caller's device buffer
int geo_encode_batch_dev_out(uvol_ctx *ctx, const uvol_mesh *meshes, int n, hipStream_t producer, uint8_t *dev_out, size_t dev_cap, size_t *out_offs, size_t *out_lens, int *status);
int geodec_create(uvol_ctx *ctx);
void geodec_destroy(uvol_ctx *ctx);
int geo_decode_batch(uvol_ctx *ctx, const uint8_t *const *files, const size_t *lens, int n, uvol_decoded_mesh *out, int *status, bool outputs_on_device = false);
int texdec_create(uvol_ctx *ctx);
void texdec_destroy(uvol_ctx *ctx);
int tex_decode_segments(uvol_ctx *ctx, const uint8_t *const *files, const size_t *lens, int n, uint8_t *const *rgba, size_t layer_cap, bool outputs_on_device, int target, int *status = nullptr);
int obj_create(uvol_ctx *ctx);
void obj_destroy(uvol_ctx *ctx);
int obj_parse_batch(uvol_ctx *ctx, const uint8_t *const *texts, const size_t *lens, int n, int slot, uvol_mesh *meshes_out, int *status);
int png_create(uvol_ctx *ctx);
void png_destroy(uvol_ctx *ctx);
int png_order_before(uvol_ctx *ctx, hipStream_t stream, const uint8_t *const *layers, size_t n_layers);
int png_wait(uvol_ctx *ctx);
int png_unfilter_batch(uvol_ctx *ctx, const uint8_t *const *raw, int n, uint32_t w, uint32_t h, int channels, int slot, const uint8_t **rgba_dev_out);
int png_ingest_batch(uvol_ctx *ctx, const uint8_t *const *raw, const size_t *zlens, int n, uint32_t w, uint32_t h, int channels, int slot, const uint8_t **rgba_dev_out);
int png_status(uvol_ctx *ctx, int slot, int *status, int n);
int uastc_create(uvol_ctx *ctx);
void uastc_destroy(uvol_ctx *ctx);
#define UASTC_PROBE_SUPERCOMPRESSED (-10)      /* a UASTC .ktx2 (DFD colour model 166) whose level data are Zstandard-supercompressed */
int uastc_ktx2_probe(const uint8_t *b, size_t n, uint32_t *W, uint32_t *H, uint32_t *L, uint64_t *lvl_off);
int uastc_zstd_info(const uint8_t *b, size_t n, uint32_t *W, uint32_t *H, uint32_t *L);      // sizes of a Zstandard-supercompressed UASTC file from its header (nothing inflated)
int tdec_file_alpha(const uint8_t *b, size_t n);          // ETC1S .ktx2: 1 = has alpha slices, 0 = opaque, < 0 = not a file the ETC1S decoder reads (tex_decode.hip)
int uastc_unzstd(const uint8_t *b, size_t n, std::vector<uint8_t> &out);      // Zstandard-supercompressed UASTC -> the equivalent scheme-0 file (needs the system's libzstd; tex_uastc.hip)
// the files of a decode / transcode call with every Zstandard-supercompressed UASTC file replaced by its inflated equivalent (kept alive here)
struct UvolUnzstd {
  std::vector<std::vector<uint8_t>> keep; std::vector<const uint8_t *> p; std::vector<size_t> l; bool any = false;
  UvolUnzstd(const uint8_t *const *files, const size_t *lens, int n) : p(files, files + n), l(lens, lens + n) {
    for (int i = 0; i < n; i++) {
      uint32_t w, h, ly; uint64_t lo;
      if (!files[i] || uastc_ktx2_probe(files[i], lens[i], &w, &h, &ly, &lo) != UASTC_PROBE_SUPERCOMPRESSED) continue;
      std::vector<uint8_t> o;
      try { if (uastc_unzstd(files[i], lens[i], o) == 0) { keep.push_back(std::move(o)); p[i] = keep.back().data(); l[i] = keep.back().size(); any = true; } }
      catch (...) { }                                    // out of host memory: the file stays supercompressed and is refused in its slot (no exception crosses the C ABI)
    }
  }
};
int tex_uastc_encode_segments(uvol_ctx *ctx, const uint8_t *const *rgba, int n_seg, int n_layers, uint32_t w, uint32_t h,
                              bool inputs_on_device, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status = nullptr);
int tex_uastc_decode_segments(uvol_ctx *ctx, const uint8_t *const *files, const size_t *lens, int n, uint8_t *const *out, size_t layer_cap, bool outputs_on_device, int target, int *status = nullptr);
int tex_create(uvol_ctx *ctx);
void tex_destroy(uvol_ctx *ctx);
int tex_encode_segments(uvol_ctx *ctx, const uint8_t *const *rgba, int n_seg, int n_layers, uint32_t w, uint32_t h,
                        bool inputs_on_device, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status = nullptr, bool defer = false);
int tex_trim(uvol_ctx *ctx);       // uvol_trim: the texture lanes' device buffers go back to the device
int tex_flush(uvol_ctx *ctx);      // completes the parts an enqueued texture call left in flight
int tex_encode_segment(uvol_ctx *ctx, const uint8_t *const *rgba, int n_layers, uint32_t w, uint32_t h,
                       bool inputs_on_device, uint8_t *out, size_t cap, size_t *out_len);

// ------------------------------------------------------------------------------------------------
// Host -> device upload of many caller-owned (pageable) arrays into ONE contiguous device region.  hipMemcpyAsync from
// pageable memory is staged by the runtime on a single thread (5 - 8 GB/s measured: a 240-frame batch of meshes + images is
// 6 GB, i.e. about a second, more than the GPU needs to encode it).  Here the region is cut into chunks; up to 8 host
// threads copy the pieces of the items that fall into a chunk into one of two pinned buffers laid out like the device region,
// and each chunk goes over in one DMA at link speed while the threads fill the other buffer.
// ------------------------------------------------------------------------------------------------
struct UvolUpItem { size_t dev_off; const void *src; size_t bytes; };     // sorted by dev_off, non-overlapping
// Contexts of one process that drive the SAME device share its host link.  When a geometry and a texture context upload at the
// same time (uvolenc, the PCIe-inclusive bench variant) each gets half of it and BOTH encoders start late; chunks are therefore
// issued shortest-remaining-upload first, so the smaller batch (the meshes) is on the device - and being encoded - while the
// larger one goes over.  One gate PER DEVICE: uploads to different GPUs travel on different links and never wait for each other
// (`uvolenc --gpus N` runs 2 contexts per GPU in one process).  A context that has been passed over UP_MAX_SKIPS times goes next
// whatever its size, so a stream of small uploads cannot starve a large one.
struct UvolUpSched {
  enum { UP_MAX_SKIPS = 64 };
  struct Ent { size_t rem; unsigned skipped; };
  std::mutex m; std::condition_variable cv; std::map<uint64_t, Ent> rem; uint64_t next_id = 1;
  uint64_t enter(size_t total) { std::lock_guard<std::mutex> l(m); const uint64_t id = next_id++; rem[id] = Ent{ total, 0 }; return id; }
  // blocks until `id` has the least bytes left (ties: the older one), or has waited through UP_MAX_SKIPS chunks of others
  void turn(uint64_t id) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] {
      const Ent &me = rem[id];
      if (me.skipped >= UP_MAX_SKIPS) { for (auto &kv : rem) if (kv.first < id && kv.second.skipped >= UP_MAX_SKIPS) return false; return true; }
      for (auto &kv : rem) if (kv.first != id && (kv.second.skipped >= UP_MAX_SKIPS || kv.second.rem < me.rem || (kv.second.rem == me.rem && kv.first < id))) return false;
      return true; });
  }
  void progress(uint64_t id, size_t bytes) {
    { std::lock_guard<std::mutex> l(m); Ent &r = rem[id]; r.rem = r.rem > bytes ? r.rem - bytes : 0; r.skipped = 0; for (auto &kv : rem) if (kv.first != id) kv.second.skipped++; }
    cv.notify_all();
  }
  void leave(uint64_t id) { { std::lock_guard<std::mutex> l(m); rem.erase(id); } cv.notify_all(); }
};
inline UvolUpSched &uvol_up_sched(int device) {
  static std::mutex m; static std::map<int, UvolUpSched *> by_dev;
  std::lock_guard<std::mutex> l(m);
  UvolUpSched *&s = by_dev[device]; if (!s) s = new UvolUpSched();      // lives as long as the process (contexts come and go)
  return *s;
}
// host threads one upload may use for its memcpy into the pinned buffers: 8 when the process drives one device, fewer per upload
// when it drives several (`uvolenc --gpus 8`: 16 contexts), so that the copies do not oversubscribe the host
inline int uvol_up_threads() {
  static const int forced = [] { const char *e = getenv("UVOL_UP_THREADS"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();      // diagnostic
  if (forced) return forced;
  static const int hw = [] { const unsigned h = std::thread::hardware_concurrency(); return (int)(h ? h : 8); }();
  int ndev = 1; (void)hipGetDeviceCount(&ndev); if (ndev < 1) ndev = 1;
  return std::max(2, std::min(8, hw / (2 * ndev)));
}
// is [p, p + n) inside a buffer handed out by uvol_host_alloc (page-locked: the DMA engines read it directly)?  (uvol_api.cpp)
bool uvol_host_pinned(const void *p, size_t n);
static inline int uvol_upload_staged(uvol_ctx *ctx, uint8_t *dev_base, const std::vector<UvolUpItem> &items) {
  if (items.empty()) return UVOL_OK;
  // A caller that keeps its arrays in uvol_host_alloc memory ("inputs resident in pinned host memory") skips the staging
  // copy: every array goes from where it lies to the device, asynchronously, in call order.  One pageable array and the whole call is staged.
  // The copies still take their turns at the device's upload gate (below), in runs of about 128 MB with at most two runs queued: queued all at
  // once, the copies of a geometry and a texture context share the link, both encoders start late and nothing overlaps the upload
  // (2560 frames + 512 segments: 1356 frames/s against 1750 through the staging buffers; profiles/r05_i_bench_host_pinned.json).
  { bool all = true; for (const UvolUpItem &it : items) if (it.bytes && !uvol_host_pinned(it.src, it.bytes)) { all = false; break; }
    if (all) {
      size_t tot = 0; for (const UvolUpItem &it : items) tot += it.bytes;
      const size_t RUN = (size_t)128 << 20;
      if (tot < 2 * RUN) { for (const UvolUpItem &it : items) if (it.bytes) UVOL_HIP_CHECK(ctx, hipMemcpyAsync(dev_base + it.dev_off, it.src, it.bytes, hipMemcpyHostToDevice, ctx->stream)); return UVOL_OK; }
      for (int k = 0; k < 2; k++) if (!ctx->pin_ev[k]) UVOL_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->pin_ev[k], hipEventDisableTiming));
      UvolUpSched &sched = uvol_up_sched(ctx->device);
      const uint64_t sid = sched.enter(tot);
      struct Leave { UvolUpSched &s; uint64_t id; ~Leave() { s.leave(id); } } leave_{ sched, sid };
      bool rec[2] = { false, false }; int buf = 0;
      for (size_t i = 0; i < items.size(); buf ^= 1) {
        sched.turn(sid);
        if (rec[buf]) { UVOL_HIP_CHECK(ctx, hipEventSynchronize(ctx->pin_ev[buf])); rec[buf] = false; }      // the run before the last one has gone over
        size_t run = 0;
        for (; i < items.size() && (run == 0 || run + items[i].bytes <= RUN); i++) {
          const UvolUpItem &it = items[i]; if (!it.bytes) continue;
          UVOL_HIP_CHECK(ctx, hipMemcpyAsync(dev_base + it.dev_off, it.src, it.bytes, hipMemcpyHostToDevice, ctx->stream)); run += it.bytes;
        }
        UVOL_HIP_CHECK(ctx, hipEventRecord(ctx->pin_ev[buf], ctx->stream)); rec[buf] = true;
        sched.progress(sid, run);
      }
      return UVOL_OK;
    } }
  const size_t total = items.back().dev_off + items.back().bytes;
  const size_t CH = (size_t)128 << 20;
  if (total < ((size_t)4 << 20)) {                                           // small batches: the runtime's own path
    for (const UvolUpItem &it : items) if (it.bytes) UVOL_HIP_CHECK(ctx, hipMemcpyAsync(dev_base + it.dev_off, it.src, it.bytes, hipMemcpyHostToDevice, ctx->stream));
    return UVOL_OK;
  }
  if (!ctx->up_pin[0]) {                                                    // both buffers and both events, or nothing: published only when all four exist
    uint8_t *pin[2] = { nullptr, nullptr }; hipEvent_t ev[2] = { nullptr, nullptr }; hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess; k++) { e = hipHostMalloc((void **)&pin[k], CH, hipHostMallocDefault); if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming); }
    if (e != hipSuccess) {
      for (int k = 0; k < 2; k++) { if (pin[k]) (void)hipHostFree(pin[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); }
      ctx->set_error("staged upload: pinned buffers: %s", hipGetErrorString(e)); return UVOL_E_HIP;
    }
    for (int k = 0; k < 2; k++) { ctx->up_pin[k] = pin[k]; ctx->up_ev[k] = ev[k]; ctx->up_rec[k] = false; }
    ctx->up_cap = CH;
  }
  // a previous call may have returned early (an error after its DMAs were queued): nothing is copied into a pinned buffer a DMA still reads
  for (int k = 0; k < 2; k++) if (ctx->up_rec[k]) { UVOL_HIP_CHECK(ctx, hipEventSynchronize(ctx->up_ev[k])); ctx->up_rec[k] = false; }
  size_t first = 0; int buf = 0;
  UvolUpSched &sched = uvol_up_sched(ctx->device);
  const uint64_t sid = sched.enter(total);
  struct Leave { UvolUpSched &s; uint64_t id; ~Leave() { s.leave(id); } } leave_{ sched, sid };
  const int nt = uvol_up_threads();
  for (size_t c0 = 0; c0 < total; c0 += CH, buf ^= 1) {
    const size_t c1 = std::min(total, c0 + CH);
    sched.turn(sid);
    if (ctx->up_rec[buf]) { UVOL_HIP_CHECK(ctx, hipEventSynchronize(ctx->up_ev[buf])); ctx->up_rec[buf] = false; }      // the DMA that last read this buffer is done
    while (first < items.size() && items[first].dev_off + items[first].bytes <= c0) first++;
    size_t last = first; while (last < items.size() && items[last].dev_off < c1) last++;
    uint8_t *pin = ctx->up_pin[buf];
    auto work = [&](size_t a, size_t b) {
      for (size_t i = a; i < b; i++) {
        const UvolUpItem &it = items[i];
        const size_t lo = std::max(it.dev_off, c0), hi = std::min(it.dev_off + it.bytes, c1);
        if (hi > lo) memcpy(pin + (lo - c0), (const uint8_t *)it.src + (lo - it.dev_off), hi - lo);
      } };
    const size_t ni = last - first;
    if (ni >= 8) { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work, first + ni * t / nt, first + ni * (t + 1) / nt); for (auto &x : th) x.join(); }
    else if (ni >= 1) {                                                       // few large items (image layers): split each across the threads
      std::vector<std::thread> th;
      for (int t = 0; t < nt; t++) th.emplace_back([&, t]() {
        for (size_t i = first; i < last; i++) {
          const UvolUpItem &it = items[i];
          const size_t lo = std::max(it.dev_off, c0), hi = std::min(it.dev_off + it.bytes, c1); if (hi <= lo) continue;
          const size_t n = hi - lo, a = n * t / nt, b = n * (t + 1) / nt;
          if (b > a) memcpy(pin + (lo - c0) + a, (const uint8_t *)it.src + (lo - it.dev_off) + a, b - a);
        } });
      for (auto &x : th) x.join();
    }
    UVOL_HIP_CHECK(ctx, hipMemcpyAsync(dev_base + c0, pin, c1 - c0, hipMemcpyHostToDevice, ctx->stream));
    UVOL_HIP_CHECK(ctx, hipEventRecord(ctx->up_ev[buf], ctx->stream));
    ctx->up_rec[buf] = true;
    sched.progress(sid, c1 - c0);
  }
  return UVOL_OK;
}

// ------------------------------------------------------------------------------------------------
// Uplink (round 6): the upload ring of a context for calls whose inputs ALL lie in uvol_host_alloc (page-locked) memory - SURVEY 8(d)'s
// boundary, "inputs resident in pinned host memory".  Such a call's uploads need no host thread: every group (geometry) / part (texture)
// of the call gets a SLOT of the ring - a device buffer, a `ready` event and a `released` event - and the copies of ALL its groups are
// queued on the context's own COPY STREAM when the call begins, before any kernel of the call is enqueued:
//     copy stream :  wait(released of the slot's previous user) -> DMAs -> record(ready)
//     lane stream :  wait(ready) -> the group's kernels -> record(released)
// so the host link is busy from the first group's first byte to the last group's last byte whatever the host thread is waiting for in
// between (a lane's previous group, the mid-batch read-back of geo_submit), and the next enqueued call's uploads queue up behind this
// call's while its kernels still run.  Until round 5 the copies went on the lane's stream from inside the group's submission, 15 k
// copies of 0.4 - 2.4 MB per 2560-frame pass in runs of 128 MB with two runs queued: the link idled whenever the host thread did
// anything else (1458 - 1614 frames/s against 1520 - 1776 through the staging buffers, profiles/r05_final3_variant_host_inputs_2560.json).
// The device layout MIRRORS the host layout (UvolUpPlacer): arrays that lie back to back in the caller's arena lie back to back in
// the slot, and go over in ONE copy per contiguous run (a whole frame, or many frames) instead of one per array - the link delivers
// 57 GB/s for copies of >= 16 MiB, 36 for 1 MiB (profiles/r05_h2d_rate.json).
// One thread per context drives its uplink (the caller's, or the context's enqueue worker); the ring is not shared between contexts.
// ------------------------------------------------------------------------------------------------
struct UvolUpChunk { const uint8_t *src; uint8_t *dst; unsigned long long bytes; };      // one workgroup-sized piece of a slot's copy list (k_uplink_copy)
struct UvolUpSlot {
  uvol_devbuf buf;
  uvol_devbuf list_dev; UvolUpChunk *list_host = nullptr; size_t list_cap = 0; bool filled = false;      // the copy list: written into page-locked memory, read by the kernel from its device copy
  hipEvent_t ready = nullptr, released = nullptr;
  bool rel_rec = false;            // `released` has been recorded behind the kernels of the slot's current content
  uint64_t gen = 0;                // bumped by every fill: a consumer that comes back later (the texture's alpha re-run) sees whether its bytes are still there
};
struct UvolUplink { hipStream_t stream = nullptr; std::vector<UvolUpSlot *> slots; size_t next = 0; };
// slots beyond one per lane (UVOL_UPLINK_AHEAD, default 2): with exactly one slot per lane a group's upload could not start before the lane's
// previous group had released its slot, and the lane idled while its inputs crossed the link; the extra slots let the link run that many
// groups ahead of the lanes (a slot of 640 frames is 6.8 GB)
static inline int uvol_uplink_ahead() { static const int v = [] { const char *e = getenv("UVOL_UPLINK_AHEAD"); const int k = e ? atoi(e) : 2; return k < 0 ? 0 : (k > 16 ? 16 : k); }(); return v; }
static inline bool uvol_uplink_enabled() { static const bool v = [] { const char *e = getenv("UVOL_UPLINK"); return !(e && *e == '0'); }(); return v; }      // UVOL_UPLINK=0 (diagnostic): round 5's in-submission copies
// device offsets that mirror the host layout: an array that starts (almost) where its predecessor ended in host memory is placed
// at the same distance behind it in the slot; anything else starts a new 256-byte-aligned run.  Every array keeps a 256-byte-aligned
// device address (the kernels' vector loads), so runs are continued by 256-byte-aligned host arrays only.
struct UvolUpPlacer {
  size_t off = 0; uintptr_t prev_end = 0; size_t prev_dev_end = 0; bool run_ok = false;
  size_t place(const void *src, size_t bytes) {
    const uintptr_t s = (uintptr_t)src; size_t d;
    if (run_ok && s >= prev_end && s - prev_end <= 4096 && (s & 255) == 0 && ((prev_dev_end + (s - prev_end)) & 255) == 0) d = prev_dev_end + (size_t)(s - prev_end);
    else { d = (off + 255) & ~(size_t)255; run_ok = (s & 255) == 0; }
    prev_end = s + bytes; prev_dev_end = d + bytes; off = prev_dev_end; return d;
  }
  size_t total() const { return (off + 255) & ~(size_t)255; }
};
static inline void uvol_uplink_destroy(uvol_ctx *ctx) {
  UvolUplink *U = ctx->uplink; if (!U) return;
  if (U->stream) { (void)hipStreamSynchronize(U->stream); }
  for (UvolUpSlot *S : U->slots) {
    if (S->rel_rec && S->released) (void)hipEventSynchronize(S->released);
    if (S->buf.p) (void)hipFree(S->buf.p);
    if (S->list_dev.p) (void)hipFree(S->list_dev.p);
    if (S->list_host) (void)hipHostFree(S->list_host);
    if (S->ready) (void)hipEventDestroy(S->ready);
    if (S->released) (void)hipEventDestroy(S->released);
    delete S;
  }
  if (U->stream) (void)hipStreamDestroy(U->stream);
  delete U; ctx->uplink = nullptr;
}
// uvol_trim: the slots' device buffers go back to the device (nothing of the context is in flight)
static inline void uvol_uplink_trim(uvol_ctx *ctx) {
  UvolUplink *U = ctx->uplink; if (!U) return;
  if (U->stream) (void)hipStreamSynchronize(U->stream);
  for (UvolUpSlot *S : U->slots) { if (S->rel_rec && S->released) (void)hipEventSynchronize(S->released); S->rel_rec = false; if (S->buf.p) { (void)hipFree(S->buf.p); S->buf.p = nullptr; S->buf.cap = 0; } }
}
// the ring with at least `n` slots (it only grows); nullptr + error text when a stream / event cannot be created
static inline UvolUplink *uvol_uplink(uvol_ctx *ctx, size_t n) {
  if (!ctx->uplink) {
    UvolUplink *U = new UvolUplink();
    if (hipStreamCreateWithFlags(&U->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); delete U; ctx->set_error("uplink: copy stream creation failed"); return nullptr; }
    ctx->uplink = U;
  }
  UvolUplink *U = ctx->uplink;
  while (U->slots.size() < n) {
    UvolUpSlot *S = new UvolUpSlot();
    if (hipEventCreateWithFlags(&S->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&S->released, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError(); if (S->ready) (void)hipEventDestroy(S->ready); delete S; ctx->set_error("uplink: event creation failed"); return nullptr; }
    U->slots.push_back(S);
  }
  return U;
}
// Fill the ring's next slot: `items` (sorted by dev_off, placed by UvolUpPlacer, every source in uvol_host_alloc memory) go to a buffer of
// `total` bytes on the copy stream, behind the release of whatever the slot held.  Returns the slot (nullptr + error text on failure).
// Nothing here waits for the device unless the slot's buffer has to grow.  (uvol_api.cpp: the copy is ONE gather kernel per slot.)
UvolUpSlot *uvol_uplink_fill(uvol_ctx *ctx, UvolUplink *U, const std::vector<UvolUpItem> &items, size_t total);
// consumer side: `stream` reads the slot from here on ...
static inline int uvol_uplink_acquire(uvol_ctx *ctx, UvolUpSlot *S, hipStream_t stream) { UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(stream, S->ready, 0)); return UVOL_OK; }
// ... and everything queued on `stream` so far was the last of it
static inline int uvol_uplink_release(uvol_ctx *ctx, UvolUpSlot *S, hipStream_t stream) { UVOL_HIP_CHECK(ctx, hipEventRecord(S->released, stream)); S->rel_rec = true; return UVOL_OK; }

// Device -> host download of many arrays into caller-owned (pageable) memory: the mirror of uvol_upload_staged.  hipMemcpyAsync into
// pageable memory is staged by the runtime on one thread (a decoded 1920-frame batch is 20 GB: several seconds); here consecutive arrays
// are copied by DMA into one of two pinned buffers, packed, and host threads copy a buffer out into the caller's arrays while the DMAs
// of the next one run.  The caller's stream is synchronised when the function returns.
struct UvolDnItem { const void *src; void *dst; size_t bytes; };
static inline int uvol_download_staged(uvol_ctx *ctx, const std::vector<UvolDnItem> &items) {
  size_t total = 0; for (const UvolDnItem &it : items) total += it.bytes;
  const size_t CH = (size_t)128 << 20;
  bool big = false; for (const UvolDnItem &it : items) big = big || it.bytes > CH;
  // Outputs that ALL lie in uvol_host_alloc memory (round 6): the DMA engines write them where the caller wants them - no pinned staging buffer,
  // no host thread copying 10.5 MB per decoded frame out of it (the staged form's limit: ~20 GB/s of memcpy into pageable memory).  Arrays that
  // are contiguous on both sides go as one copy.
  { bool all = !items.empty(); for (const UvolDnItem &it : items) if (it.bytes && !uvol_host_pinned(it.dst, it.bytes)) { all = false; break; }
    if (all) {
      for (size_t i = 0; i < items.size();) {
        size_t len = items[i].bytes, j = i + 1;
        for (; j < items.size(); j++) { if ((const uint8_t *)items[j].src != (const uint8_t *)items[i].src + len || (uint8_t *)items[j].dst != (uint8_t *)items[i].dst + len || len + items[j].bytes > ((size_t)512 << 20) || !uvol_host_pinned(items[i].dst, len + items[j].bytes)) break; len += items[j].bytes; }
        if (len) UVOL_HIP_CHECK(ctx, hipMemcpyAsync(items[i].dst, items[i].src, len, hipMemcpyDeviceToHost, ctx->stream));
        i = j;
      }
      UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      return UVOL_OK;
    } }
  if (total < ((size_t)4 << 20) || big) {                                  // small calls (or one huge array): the runtime's own path
    for (const UvolDnItem &it : items) if (it.bytes) UVOL_HIP_CHECK(ctx, hipMemcpyAsync(it.dst, it.src, it.bytes, hipMemcpyDeviceToHost, ctx->stream));
    UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return UVOL_OK;
  }
  if (!ctx->dn_pin[0]) {
    uint8_t *pin[2] = { nullptr, nullptr }; hipEvent_t ev[2] = { nullptr, nullptr }; hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess; k++) { e = hipHostMalloc((void **)&pin[k], CH, hipHostMallocDefault); if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming); }
    if (e != hipSuccess) {
      for (int k = 0; k < 2; k++) { if (pin[k]) (void)hipHostFree(pin[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); }
      ctx->set_error("staged download: pinned buffers: %s", hipGetErrorString(e)); return UVOL_E_HIP;
    }
    for (int k = 0; k < 2; k++) { ctx->dn_pin[k] = pin[k]; ctx->dn_ev[k] = ev[k]; }
  }
  const int nt = uvol_up_threads();
  size_t ra[2] = { 0, 0 }, rb[2] = { 0, 0 }; bool has[2] = { false, false };
  auto drain = [&](int b) -> int {                                          // buffer b holds items [ra[b], rb[b]) packed: out to the caller's arrays
    UVOL_HIP_CHECK(ctx, hipEventSynchronize(ctx->dn_ev[b]));
    const size_t a = ra[b], e = rb[b];
    std::vector<size_t> off(e - a + 1, 0); for (size_t i = a; i < e; i++) off[i - a + 1] = off[i - a] + items[i].bytes;
    const size_t bytes = off[e - a]; const uint8_t *pin = ctx->dn_pin[b];
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&, t]() {                   // thread t takes bytes [lo, hi) of the packed buffer, whatever arrays they belong to
      const size_t lo = bytes * (size_t)t / (size_t)nt, hi = bytes * (size_t)(t + 1) / (size_t)nt;
      for (size_t i = a; i < e; i++) {
        const size_t s0 = std::max(off[i - a], lo), s1 = std::min(off[i - a + 1], hi);
        if (s1 > s0) memcpy((uint8_t *)items[i].dst + (s0 - off[i - a]), pin + s0, s1 - s0);
      } });
    for (auto &x : th) x.join();
    has[b] = false;
    return UVOL_OK; };
  size_t i = 0; int buf = 0;
  while (i < items.size()) {
    if (has[buf]) { const int r = drain(buf); if (r != UVOL_OK) return r; }
    const size_t a = i; size_t o = 0;
    while (i < items.size() && o + items[i].bytes <= CH) {
      if (items[i].bytes) UVOL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->dn_pin[buf] + o, items[i].src, items[i].bytes, hipMemcpyDeviceToHost, ctx->stream));
      o += items[i].bytes; i++;
    }
    UVOL_HIP_CHECK(ctx, hipEventRecord(ctx->dn_ev[buf], ctx->stream));
    ra[buf] = a; rb[buf] = i; has[buf] = true;
    buf ^= 1;
  }
  for (int k = 0; k < 2; k++, buf ^= 1) if (has[buf]) { const int r = drain(buf); if (r != UVOL_OK) return r; }      // (the older buffer first)
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return UVOL_OK;
}
