// uvol_api.cpp — the extern "C" surface declared in include/uvol_codec.h.
// Each entry point replaces a process boundary of scripts/Encoder.py (:260-262 draco_encoder,
// :290-292 basisu); see the header for the mapping.  No CPU fallback: without a HIP device
// uvol_ctx_create returns UVOL_E_NODEVICE.
#include "uvol_common.hpp"
#include <new>

// Optional CU partition (params.cu_mod / cu_residues): the context's streams only run on CUs whose index modulo
// cu_mod has its bit set in cu_residues.  bench.py --cu-split gives geometry 3 of every 4 CUs and texture the 4th,
// so the latency-bound one-wave walkers never share a CU's memory pipeline with the streaming texture kernels.
// Page-locked host memory for callers that keep their inputs in it (uvol_host_alloc / uvol_host_free): the registry uvol_upload_staged asks.
namespace { std::mutex g_pin_m; std::map<uintptr_t, size_t> g_pin; }
bool uvol_host_pinned(const void *p, size_t n) {
  std::lock_guard<std::mutex> l(g_pin_m);
  if (g_pin.empty()) return false;
  auto it = g_pin.upper_bound((uintptr_t)p);
  if (it == g_pin.begin()) return false;
  --it;
  return (uintptr_t)p >= it->first && (uintptr_t)p + n <= it->first + it->second;
}
extern "C" void *uvol_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> l(g_pin_m); g_pin[(uintptr_t)p] = bytes; return p;
}
extern "C" void uvol_host_free(void *p) {
  if (!p) return;
  { std::lock_guard<std::mutex> l(g_pin_m); g_pin.erase((uintptr_t)p); }
  (void)hipHostFree(p);
}
// ------------------------------------------------------------------------------------------------
// Uplink copy (uvol_common.hpp "Uplink").  Page-locked host memory is mapped into the device's address space, so the upload of a slot is
// ONE kernel on the context's copy stream that reads the caller's arrays over the link and writes them where the encoders expect them:
// a list of chunks of <= 256 KiB, one workgroup per chunk (grid-stride), 16-byte loads with eight in flight per lane.  A few dozen
// workgroups keep the link full (its bandwidth-delay product is ~100 KB); they spend their time waiting, not issuing.
// Measured (round 6, profiles/r06_uplink_forms.json) against the runtime's own copies on the same copy stream: hipMemcpyAsync of the
// >= 84 MB runs of a mirrored layout goes to the SDMA engines, which delivered 22 GB/s per stream beside the encoders' kernels (36 GB/s for
// both contexts together, 1295 frames/s); copies of <= 2.4 MB go through the runtime's blit kernel, one launch per array (1604).
// UVOL_UPLINK_DMA=1 (diagnostic) keeps the runtime's copies.
// ------------------------------------------------------------------------------------------------
#define UPLINK_CHUNK ((size_t)256 << 10)
__global__ void __launch_bounds__(256) k_uplink_copy(const UvolUpChunk *list, unsigned n_chunks) {
  for (unsigned c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const UvolUpChunk ch = list[c];
    const uint8_t *src = ch.src; uint8_t *dst = ch.dst; const size_t n = (size_t)ch.bytes;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
      const uint4 *s4 = (const uint4 *)src; uint4 *d4 = (uint4 *)dst; const size_t n4 = n >> 4;
      size_t i = threadIdx.x;
      for (; i + 7 * 256 < n4; i += 8 * 256) {              // eight independent 16-byte reads over the link per lane
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = s4[i + (size_t)k * 256];
#pragma unroll
        for (int k = 0; k < 8; k++) d4[i + (size_t)k * 256] = v[k];
      }
      for (; i < n4; i += 256) d4[i] = s4[i];
      for (size_t b = (n4 << 4) + threadIdx.x; b < n; b += 256) dst[b] = src[b];
    } else if ((((uintptr_t)src | (uintptr_t)dst) & 3) == 0) {
      const uint32_t *s1 = (const uint32_t *)src; uint32_t *d1 = (uint32_t *)dst; const size_t n1 = n >> 2;
      for (size_t i = threadIdx.x; i < n1; i += 256) d1[i] = s1[i];
      for (size_t b = (n1 << 2) + threadIdx.x; b < n; b += 256) dst[b] = src[b];
    } else {
      for (size_t b = threadIdx.x; b < n; b += 256) dst[b] = src[b];
    }
  }
}
static inline bool uplink_dma() { static const bool v = [] { const char *e = getenv("UVOL_UPLINK_KERNEL"); return !(e && *e == '1'); }(); return v; }      // UVOL_UPLINK_KERNEL=1 (diagnostic): the gather kernel instead of the runtime's copies
static inline unsigned uplink_wgs() { static const unsigned v = [] { const char *e = getenv("UVOL_UPLINK_WGS"); const int k = e ? atoi(e) : 0; return (unsigned)(k >= 1 && k <= 4096 ? k : 64); }(); return v; }
UvolUpSlot *uvol_uplink_fill(uvol_ctx *ctx, UvolUplink *U, const std::vector<UvolUpItem> &items, size_t total) {
  UvolUpSlot *S = U->slots[U->next % U->slots.size()]; U->next++;
  if (total > S->buf.cap) {                                 // (re)allocation: the slot's last consumer first
    if (S->rel_rec) { if (hipEventSynchronize(S->released) != hipSuccess) { ctx->set_error("uplink: waiting for a slot failed"); return nullptr; } S->rel_rec = false; }
    if (hipStreamSynchronize(U->stream) != hipSuccess) { ctx->set_error("uplink: copy stream failed"); return nullptr; }
    if (S->buf.p) { (void)hipFree(S->buf.p); S->buf.p = nullptr; S->buf.cap = 0; }
    const size_t want = total + total / 16 + 4096;
    if (hipMalloc(&S->buf.p, want) != hipSuccess) { (void)hipGetLastError(); S->buf.p = nullptr; ctx->set_error("uplink: %zu bytes of device memory for a slot", want); return nullptr; }
    S->buf.cap = want;
  }
  // the slot's copy list lives in page-locked memory until its device copy has been made: the previous fill of this slot has long run
  // (its consumers' kernels have been enqueued behind it), the wait is a formality
  if (S->filled && hipEventSynchronize(S->ready) != hipSuccess) { ctx->set_error("uplink: waiting for a slot's previous fill failed"); return nullptr; }
  if (S->rel_rec) { if (hipStreamWaitEvent(U->stream, S->released, 0) != hipSuccess) { ctx->set_error("uplink: hipStreamWaitEvent failed"); return nullptr; } S->rel_rec = false; }
  S->gen++;
  if (uplink_dma()) {
    // consecutive items that are contiguous on both sides (gaps of the caller's alignment padding included) travel as one copy of up to 512 MiB
    const size_t MAXC = (size_t)512 << 20;
    for (size_t i = 0; i < items.size();) {
      const UvolUpItem &a = items[i]; size_t len = a.bytes, j = i + 1;
      for (; j < items.size(); j++) {
        const UvolUpItem &b = items[j];
        const uintptr_t ha = (uintptr_t)a.src + len, hb = (uintptr_t)b.src;
        if (hb < ha || hb - ha > 4096 || b.dev_off != a.dev_off + len + (size_t)(hb - ha) || len + (hb - ha) + b.bytes > MAXC) break;
        if (!uvol_host_pinned(a.src, len + (size_t)(hb - ha) + b.bytes)) break;      // one page-locked allocation holds both (and the padding between them)
        len += (size_t)(hb - ha) + b.bytes;
      }
      if (len && hipMemcpyAsync((uint8_t *)S->buf.p + a.dev_off, a.src, len, hipMemcpyHostToDevice, U->stream) != hipSuccess) { ctx->set_error("uplink: hipMemcpyAsync failed"); return nullptr; }
      i = j;
    }
  } else {
    size_t nch = 0; for (const UvolUpItem &it : items) nch += (it.bytes + UPLINK_CHUNK - 1) / UPLINK_CHUNK;
    if (nch > 0xffffffffull) { ctx->set_error("uplink: copy list too long"); return nullptr; }
    if (nch > S->list_cap) {
      if (S->list_host) { (void)hipHostFree(S->list_host); S->list_host = nullptr; } S->list_cap = 0;
      if (S->list_dev.p) { (void)hipFree(S->list_dev.p); S->list_dev.p = nullptr; S->list_dev.cap = 0; }
      const size_t cap = nch + nch / 4 + 64;
      if (hipHostMalloc((void **)&S->list_host, cap * sizeof(UvolUpChunk), hipHostMallocDefault) != hipSuccess || hipMalloc(&S->list_dev.p, cap * sizeof(UvolUpChunk)) != hipSuccess) {
        (void)hipGetLastError(); if (S->list_host) { (void)hipHostFree(S->list_host); S->list_host = nullptr; } ctx->set_error("uplink: copy list allocation failed"); return nullptr; }
      S->list_cap = cap; S->list_dev.cap = cap * sizeof(UvolUpChunk);
    }
    size_t c = 0;
    for (const UvolUpItem &it : items) for (size_t o = 0; o < it.bytes; o += UPLINK_CHUNK)
      S->list_host[c++] = UvolUpChunk{ (const uint8_t *)it.src + o, (uint8_t *)S->buf.p + it.dev_off + o, (unsigned long long)std::min(UPLINK_CHUNK, it.bytes - o) };
    if (nch) {
      if (hipMemcpyAsync(S->list_dev.p, S->list_host, nch * sizeof(UvolUpChunk), hipMemcpyHostToDevice, U->stream) != hipSuccess) { ctx->set_error("uplink: copy list upload failed"); return nullptr; }
      hipLaunchKernelGGL(k_uplink_copy, dim3((unsigned)std::min<size_t>(nch, uplink_wgs())), dim3(256), 0, U->stream, (const UvolUpChunk *)S->list_dev.p, (unsigned)nch);
      if (hipGetLastError() != hipSuccess) { ctx->set_error("uplink: copy kernel launch failed"); return nullptr; }
    }
  }
  if (hipEventRecord(S->ready, U->stream) != hipSuccess) { ctx->set_error("uplink: hipEventRecord failed"); return nullptr; }
  S->filled = true;
  return S;
}

hipError_t uvol_make_stream(uvol_ctx *ctx, hipStream_t *out) {
#ifndef HIPEMU
  // cu_mod == -1: the CUs with per-XCD ordinal in [lo, hi), cu_residues = lo << 8 | hi.  Mask bit i is CU ordinal i / 8 of XCD i % 8
  // (profiles/r05_xcd_census.json; an XCD whose bits are all clear is NOT excluded - its mask then counts as "all CUs" and the dispatcher
  // deals workgroups round-robin over the eight XCDs whatever the mask - so a queue cannot be kept off an XCD, only off CUs inside each one)
  if (ctx->prm.cu_mod == -1 && ctx->prm.cu_residues != 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) {
      const int ncu = prop.multiProcessorCount, lo = (ctx->prm.cu_residues >> 8) & 255, hi = ctx->prm.cu_residues & 255; std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
      for (int i = 0; i < ncu; i++) if (i / 8 >= lo && i / 8 < hi) mask[(size_t)i / 32] |= 1u << (i % 32);
      const hipError_t e = hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data());
      if (uvol_debug()) fprintf(stderr, "[uvol] CU-masked stream: per-XCD CU ordinals [%d, %d) -> %s\n", lo, hi, hipGetErrorString(e));
      if (e == hipSuccess) return hipSuccess;
      (void)hipGetLastError();
    }
  }
  if (ctx->prm.cu_mod > 1 && ctx->prm.cu_mod <= 32 && ctx->prm.cu_residues != 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) {
      const int ncu = prop.multiProcessorCount; std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
      for (int i = 0; i < ncu; i++) if ((ctx->prm.cu_residues >> (i % ctx->prm.cu_mod)) & 1) mask[(size_t)i / 32] |= 1u << (i % 32);
      const hipError_t e = hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data());
      if (uvol_debug()) fprintf(stderr, "[uvol] CU-masked stream: %d CUs, mod %d residues 0x%x -> %s\n", ncu, ctx->prm.cu_mod, ctx->prm.cu_residues, hipGetErrorString(e));
      if (e == hipSuccess) return hipSuccess;
      (void)hipGetLastError();
    }
  }
#endif
#ifndef HIPEMU
  if (ctx->prm.stream_priority > 0) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
  }
#endif
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

// ------------------------------------------------------------------------------------------------
// Enqueue form of the ABI (SURVEY 8(b) "Threading": ABI calls enqueue on the ctx, uvol_sync(ctx) completes).  A call returns as
// soon as its arguments are recorded; the work runs on the context's worker thread in call order, through the same code as the
// blocking entry points.  Input arrays, output buffers and the out_lens / status arrays belong to the call until uvol_sync(ctx)
// returns; the pointer / capacity ARRAYS themselves are copied.  The first failing call's return code and message are kept and
// returned by uvol_sync (then cleared).  A blocking entry point on a context with queued work waits for it first.
// ------------------------------------------------------------------------------------------------
static void async_worker(uvol_ctx *ctx) {
  uvol_ctx::AsyncQ *A = ctx->async;
  std::unique_lock<std::mutex> l(A->m);
  for (;;) {
    A->cv_work.wait(l, [&] { return A->stop || !A->q.empty(); });
    if (A->q.empty()) { if (A->stop) return; continue; }
    std::function<int()> f = std::move(A->q.front()); A->q.pop_front(); A->busy = true;
    l.unlock();
    int rc = UVOL_OK;
    try { rc = f(); } catch (...) { rc = UVOL_E_HIP; ctx->set_error("enqueued call: out of memory on the host"); }
    l.lock();
    if (rc != UVOL_OK && A->first_err == UVOL_OK) { A->first_err = rc; snprintf(A->err, sizeof A->err, "%s", ctx->err); }
    if (A->q.empty()) {                                    // nothing else queued: the mesh groups still in flight on the lanes are completed now
      l.unlock();
      (void)hipSetDevice(ctx->device);
      int rf = UVOL_OK;
      try { rf = geo_flush(ctx); const int rt = tex_flush(ctx); if (rf == UVOL_OK) rf = rt; } catch (...) { rf = UVOL_E_HIP; ctx->set_error("enqueued call: out of memory on the host"); }
      l.lock();
      if (rf != UVOL_OK && A->first_err == UVOL_OK) { A->first_err = rf; snprintf(A->err, sizeof A->err, "%s", ctx->err); }
    }
    A->busy = false;
    if (A->q.empty()) A->cv_idle.notify_all();
  }
}
static int async_push(uvol_ctx *ctx, std::function<int()> f) {
  try {
    if (!ctx->async) {
      // published before the thread starts (the worker reads ctx->async) and taken back if the thread cannot be started: a context
      // whose worker never ran would queue calls nobody executes and wait for them for ever in uvol_sync / uvol_ctx_destroy
      uvol_ctx::AsyncQ *A = new (std::nothrow) uvol_ctx::AsyncQ(); if (!A) return UVOL_E_HIP;
      ctx->async = A;
      try { A->th = std::thread(async_worker, ctx); } catch (...) { ctx->async = nullptr; delete A; ctx->set_error("enqueue: the worker thread could not be started"); return UVOL_E_HIP; }
    }
    { std::lock_guard<std::mutex> l(ctx->async->m); ctx->async->q.push_back(std::move(f)); }
  } catch (...) { ctx->set_error("enqueue: out of memory on the host"); return UVOL_E_HIP; }
  ctx->async->cv_work.notify_one();
  return UVOL_OK;
}
// waits for the queued calls; returns (and clears) the first error among them
static int async_drain(uvol_ctx *ctx, bool take_error = true) {
  if (!ctx->async) return UVOL_OK;
  uvol_ctx::AsyncQ *A = ctx->async;
  std::unique_lock<std::mutex> l(A->m);
  A->cv_idle.wait(l, [&] { return A->q.empty() && !A->busy; });
  if (!take_error) return UVOL_OK;
  const int rc = A->first_err;
  if (rc != UVOL_OK) { snprintf(ctx->err, sizeof ctx->err, "%s", A->err); A->first_err = UVOL_OK; A->err[0] = 0; }
  return rc;
}
// a blocking entry point on a context with queued work: the queued calls go first (their error, if any, stays for uvol_sync)
#define UVOL_AFTER_ASYNC(ctx) do { if ((ctx) && (ctx)->async) (void)async_drain((ctx), false); } while (0)
static void async_shutdown(uvol_ctx *ctx) {
  if (!ctx->async) return;
  (void)async_drain(ctx);
  { std::lock_guard<std::mutex> l(ctx->async->m); ctx->async->stop = true; }
  ctx->async->cv_work.notify_all();
  if (ctx->async->th.joinable()) ctx->async->th.join();
  delete ctx->async; ctx->async = nullptr;
}

extern "C" {

void uvol_params_default(uvol_params *p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->q_position_attr = 11; p->q_texture_attr = 10; p->q_normal_attr = 8; p->q_generic_attr = 8;
  p->draco_compression_level = 7; p->ktx2_batch_size = 5; p->etc1s_quality = 128; p->y_flip = 1; p->max_batch = 32;
}

int uvol_abi_version(void) { return UVOL_ABI_VERSION; }

int uvol_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int uvol_ctx_create(int device, const uvol_params *params, uvol_ctx **out) {
  if (!out) return UVOL_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return UVOL_E_NODEVICE;
  if (device < 0 || device >= n) return UVOL_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return UVOL_E_HIP;
  uvol_ctx *ctx = new (std::nothrow) uvol_ctx();
  if (!ctx) return UVOL_E_HIP;
  ctx->device = device;
  if (params) ctx->prm = *params; else uvol_params_default(&ctx->prm);
  if (ctx->prm.max_batch <= 0) ctx->prm.max_batch = 32;
  if (ctx->prm.etc1s_quality <= 0) ctx->prm.etc1s_quality = 128;
  if (uvol_make_stream(ctx, &ctx->stream) != hipSuccess) { delete ctx; return UVOL_E_HIP; }
  if (geo_create(ctx) != UVOL_OK || tex_create(ctx) != UVOL_OK || texdec_create(ctx) != UVOL_OK || geodec_create(ctx) != UVOL_OK || uastc_create(ctx) != UVOL_OK || obj_create(ctx) != UVOL_OK || png_create(ctx) != UVOL_OK) { uvol_ctx_destroy(ctx); return UVOL_E_HIP; }
  *out = ctx;
  return UVOL_OK;
}

void uvol_ctx_destroy(uvol_ctx *ctx) {
  if (!ctx) return;
  async_shutdown(ctx);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  ctx->resolve_profile();
  geo_destroy(ctx); tex_destroy(ctx); uvol_uplink_destroy(ctx);      // (the lanes' streams are synchronised and gone before the slots they read are freed)
  texdec_destroy(ctx); geodec_destroy(ctx); uastc_destroy(ctx); obj_destroy(ctx); png_destroy(ctx);
  for (int k = 0; k < 2; k++) { if (ctx->up_pin[k]) (void)hipHostFree(ctx->up_pin[k]); if (ctx->up_ev[k]) (void)hipEventDestroy(ctx->up_ev[k]); }
  for (int k = 0; k < 2; k++) if (ctx->pin_ev[k]) (void)hipEventDestroy(ctx->pin_ev[k]);
  for (int k = 0; k < 2; k++) { if (ctx->dn_pin[k]) (void)hipHostFree(ctx->dn_pin[k]); if (ctx->dn_ev[k]) (void)hipEventDestroy(ctx->dn_ev[k]); }
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *uvol_last_error(const uvol_ctx *ctx) { return ctx ? ctx->err : "null ctx"; }

int uvol_sync(uvol_ctx *ctx) {
  if (!ctx) return UVOL_E_INVALID;
  const int arc = async_drain(ctx);                        // every call enqueued with uvol_*_async has completed; first error among them
  (void)hipSetDevice(ctx->device);
  const int pr = png_wait(ctx);                            // ... and the last un-filter call (its layers may be read by the caller from here on)
  const hipError_t se = hipStreamSynchronize(ctx->stream);  // (the stream is synchronised whatever the two above returned)
  ctx->resolve_profile();
  if (arc != UVOL_OK) return arc;                          // the first error of the enqueued calls goes first
  if (pr != UVOL_OK) return pr;
  if (se != hipSuccess) { ctx->set_error("hipStreamSynchronize: %s", hipGetErrorString(se)); return UVOL_E_HIP; }
  return UVOL_OK;
}

int uvol_trim(uvol_ctx *ctx) {
  const int rc = uvol_sync(ctx);
  if (!ctx) return rc;
  const int rf = geo_flush(ctx);
  const int rx = tex_flush(ctx);
  int rt = geo_trim(ctx);
  const int ru = tex_trim(ctx); if (rt == UVOL_OK) rt = ru;
  uvol_uplink_trim(ctx);
  return rc != UVOL_OK ? rc : (rf != UVOL_OK ? rf : (rx != UVOL_OK ? rx : rt));
}

// defer = the enqueue form: the call's groups are submitted and completed lazily (by the worker when its queue runs empty, or when a
// later call needs the lane), so that consecutive enqueued calls overlap on the device
static int encode_batch_common(uvol_ctx *ctx, const uvol_mesh *meshes, int n, bool dev,
                               uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status, bool defer = false) {
  if (!ctx || !meshes || n < 0 || !outs || !caps || !out_lens) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  const int mb = ctx->prm.max_batch;
  for (int b0 = 0; b0 < n; b0 += mb) {
    const int nb = n - b0 < mb ? n - b0 : mb;
    int rc = defer ? geo_encode_batch_begin(ctx, meshes + b0, nb, dev, outs + b0, caps + b0, out_lens + b0, status ? status + b0 : nullptr, true)
                   : geo_encode_batch(ctx, meshes + b0, nb, dev, outs + b0, caps + b0, out_lens + b0, status ? status + b0 : nullptr);
    if (rc != UVOL_OK) return rc;
  }
  return UVOL_OK;
}

int uvol_encode_mesh(uvol_ctx *ctx, const uvol_mesh *mesh, uint8_t *out, size_t cap, size_t *out_len) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !mesh || !out || !out_len) return UVOL_E_INVALID;
  int st = 0;
  int rc = encode_batch_common(ctx, mesh, 1, false, &out, &cap, out_len, &st);
  return rc != UVOL_OK ? rc : st;
}

int uvol_encode_mesh_batch(uvol_ctx *ctx, const uvol_mesh *meshes, int n, uint8_t *const *outs, const size_t *caps,
                           size_t *out_lens, int *status) {
  UVOL_AFTER_ASYNC(ctx);
  return encode_batch_common(ctx, meshes, n, false, outs, caps, out_lens, status);
}

int uvol_encode_mesh_batch_dev(uvol_ctx *ctx, const uvol_mesh *meshes, int n, uint8_t *const *outs, const size_t *caps,
                               size_t *out_lens, int *status) {
  UVOL_AFTER_ASYNC(ctx);
  return encode_batch_common(ctx, meshes, n, true, outs, caps, out_lens, status);
}

int uvol_encode_texture_segment(uvol_ctx *ctx, const uint8_t *const *rgba, int n_layers, uint32_t width, uint32_t height,
                                uint8_t *out, size_t cap, size_t *out_len) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !rgba || n_layers <= 0 || !out || !out_len || width == 0 || height == 0) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  if (ctx->prm.uastc) return tex_uastc_encode_segments(ctx, rgba, 1, n_layers, width, height, false, &out, &cap, out_len);
  return tex_encode_segment(ctx, rgba, n_layers, width, height, false, out, cap, out_len);
}

int uvol_encode_texture_segment_dev(uvol_ctx *ctx, const uint8_t *const *rgba_dev, int n_layers, uint32_t width, uint32_t height,
                                    uint8_t *out, size_t cap, size_t *out_len) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !rgba_dev || n_layers <= 0 || !out || !out_len || width == 0 || height == 0) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  if (ctx->prm.uastc) return tex_uastc_encode_segments(ctx, rgba_dev, 1, n_layers, width, height, true, &out, &cap, out_len);
  return tex_encode_segment(ctx, rgba_dev, n_layers, width, height, true, out, cap, out_len);
}

int uvol_encode_texture_segments(uvol_ctx *ctx, const uint8_t *const *rgba, int n_segments, int n_layers, uint32_t width, uint32_t height,
                                 uint8_t *const *outs, const size_t *caps, size_t *out_lens) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !rgba || n_segments <= 0 || n_layers <= 0 || !outs || !caps || !out_lens || width == 0 || height == 0) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  if (ctx->prm.uastc) return tex_uastc_encode_segments(ctx, rgba, n_segments, n_layers, width, height, false, outs, caps, out_lens);
  return tex_encode_segments(ctx, rgba, n_segments, n_layers, width, height, false, outs, caps, out_lens);
}
int uvol_encode_texture_segments_dev(uvol_ctx *ctx, const uint8_t *const *rgba_dev, int n_segments, int n_layers, uint32_t width, uint32_t height,
                                     uint8_t *const *outs, const size_t *caps, size_t *out_lens) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !rgba_dev || n_segments <= 0 || n_layers <= 0 || !outs || !caps || !out_lens || width == 0 || height == 0) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  if (ctx->prm.uastc) return tex_uastc_encode_segments(ctx, rgba_dev, n_segments, n_layers, width, height, true, outs, caps, out_lens);
  return tex_encode_segments(ctx, rgba_dev, n_segments, n_layers, width, height, true, outs, caps, out_lens);
}

// ---- enqueue forms (see the block comment above async_worker) ----
static int mesh_batch_async(uvol_ctx *ctx, const uvol_mesh *meshes, int n, bool dev, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status) {
  if (!ctx || !meshes || n < 0 || !outs || !caps || !out_lens) return UVOL_E_INVALID;
  try {
    std::vector<uvol_mesh> m(meshes, meshes + n); std::vector<uint8_t *> o(outs, outs + n); std::vector<size_t> c(caps, caps + n);
    return async_push(ctx, [ctx, m = std::move(m), o = std::move(o), c = std::move(c), n, dev, out_lens, status]() {
      return encode_batch_common(ctx, m.data(), n, dev, o.data(), c.data(), out_lens, status, true); });
  } catch (...) { ctx->set_error("enqueue: out of memory on the host"); return UVOL_E_HIP; }
}
int uvol_encode_mesh_batch_async(uvol_ctx *ctx, const uvol_mesh *meshes, int n, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status) {
  return mesh_batch_async(ctx, meshes, n, false, outs, caps, out_lens, status);
}
int uvol_encode_mesh_batch_dev_async(uvol_ctx *ctx, const uvol_mesh *meshes, int n, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status) {
  return mesh_batch_async(ctx, meshes, n, true, outs, caps, out_lens, status);
}
static int tex_segments_async(uvol_ctx *ctx, const uint8_t *const *rgba, int n_segments, int n_layers, uint32_t width, uint32_t height, bool dev,
                              uint8_t *const *outs, const size_t *caps, size_t *out_lens) {
  if (!ctx || !rgba || n_segments <= 0 || n_layers <= 0 || !outs || !caps || !out_lens || width == 0 || height == 0) return UVOL_E_INVALID;
  try {
    std::vector<const uint8_t *> r(rgba, rgba + (size_t)n_segments * n_layers); std::vector<uint8_t *> o(outs, outs + n_segments); std::vector<size_t> c(caps, caps + n_segments);
    return async_push(ctx, [ctx, r = std::move(r), o = std::move(o), c = std::move(c), n_segments, n_layers, width, height, dev, out_lens]() {
      (void)hipSetDevice(ctx->device);
      if (ctx->prm.uastc) return tex_uastc_encode_segments(ctx, r.data(), n_segments, n_layers, width, height, dev, o.data(), c.data(), out_lens);
      return tex_encode_segments(ctx, r.data(), n_segments, n_layers, width, height, dev, o.data(), c.data(), out_lens, nullptr, true); });
  } catch (...) { ctx->set_error("enqueue: out of memory on the host"); return UVOL_E_HIP; }
}
int uvol_encode_texture_segments_async(uvol_ctx *ctx, const uint8_t *const *rgba, int n_segments, int n_layers, uint32_t width, uint32_t height,
                                       uint8_t *const *outs, const size_t *caps, size_t *out_lens) {
  return tex_segments_async(ctx, rgba, n_segments, n_layers, width, height, false, outs, caps, out_lens);
}
int uvol_encode_texture_segments_dev_async(uvol_ctx *ctx, const uint8_t *const *rgba_dev, int n_segments, int n_layers, uint32_t width, uint32_t height,
                                           uint8_t *const *outs, const size_t *caps, size_t *out_lens) {
  return tex_segments_async(ctx, rgba_dev, n_segments, n_layers, width, height, true, outs, caps, out_lens);
}

// UASTC and ETC1S files are told apart by the container (DFD colour model 166 vs 163)
static bool is_uastc(const uint8_t *const *ktx2, const size_t *lens) { uint32_t w, h, l; uint64_t lo; return uastc_ktx2_probe(ktx2[0], lens[0], &w, &h, &l, &lo) == 0; }
static int decode_dispatch(uvol_ctx *ctx, const uint8_t *const *ktx2_in, const size_t *lens_in, int n, uint8_t *const *out, size_t layer_cap, bool dev, int target) {
  // Zstandard-supercompressed UASTC files (the default of stock `basisu -uastc -ktx2`) are inflated on the host through the system's libzstd
  const UvolUnzstd Z(ktx2_in, lens_in, n);
  const uint8_t *const *ktx2 = Z.p.data(); const size_t *lens = Z.l.data();
  { uint32_t w, h, l; uint64_t lo;
    if (uastc_ktx2_probe(ktx2[0], lens[0], &w, &h, &l, &lo) == UASTC_PROBE_SUPERCOMPRESSED) {
      ctx->set_error("supercompressed UASTC (supercompressionScheme != 0): Zstandard level data are read when libzstd.so.1 is installed and the frame is intact - it is not, or this is another scheme; write the file with -ktx2_no_zstandard"); return UVOL_E_UNSUPPORTED; } }
  if (is_uastc(ktx2, lens)) {
    if (target < 0 || target > UVOL_TARGET_BC3) { ctx->set_error("unknown transcode target"); return UVOL_E_UNSUPPORTED; }      // (UASTC sources take every target since round 5)
    return tex_uastc_decode_segments(ctx, ktx2, lens, n, out, layer_cap, dev, target);
  }
  if (target == 3) { ctx->set_error("ASTC 4x4 is the transcode target of UASTC sources (KTX2Loader.js:591-600); this file is ETC1S"); return UVOL_E_UNSUPPORTED; }
  return tex_decode_segments(ctx, ktx2, lens, n, out, layer_cap, dev, target);
}
// Per-segment results and mixed batches (reference scripts/Encoder.py:293-298 fails one basisu process, not the run; SURVEY 5
// "a failed frame must not poison the batch").  Every file is looked at on its own: not a container this decoder reads -> its status; the
// first readable file fixes the batch's width / height / layer count and a file of another shape is UVOL_E_INVALID in its slot; a source
// kind that does not transcode to the target is UVOL_E_UNSUPPORTED in its slot; ETC1S and UASTC files run as one batch per kind; a file
// whose payload turns out corrupt on the device fails alone.  The call itself fails only for bad arguments, memory or the device.
int uvol_transcode_texture_segments_st(uvol_ctx *ctx, const uint8_t *const *ktx2_in, const size_t *lens_in, int n_segments,
                                       uint8_t *const *out, size_t layer_cap, int outputs_on_device, int target, int *status) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !ktx2_in || !lens_in || n_segments <= 0 || !out || !status || target < UVOL_TARGET_RGBA32 || target > UVOL_TARGET_BC3) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  const int n = n_segments;
  const UvolUnzstd Z(ktx2_in, lens_in, n);                  // (Zstandard-supercompressed UASTC files: inflated on the host, see decode_dispatch)
  const uint8_t *const *ktx2 = Z.p.data(); const size_t *lens = Z.l.data();
  std::vector<int> kind((size_t)n, -1);                    // 0 ETC1S opaque, 1 UASTC, 2 ETC1S with alpha slices
  uint32_t W0 = 0, H0 = 0, L0 = 0; bool have = false;
  for (int i = 0; i < n; i++) {
    status[i] = UVOL_E_INVALID;
    if (!ktx2[i]) continue;
    uint32_t w = 0, h = 0, l = 0; uint64_t lo = 0;
    const int pu = uastc_ktx2_probe(ktx2[i], lens[i], &w, &h, &l, &lo);
    if (pu == UASTC_PROBE_SUPERCOMPRESSED) { status[i] = UVOL_E_UNSUPPORTED; continue; }
    int k = -1;
    if (pu == 0) k = 1;
    else { const int ri = uvol_ktx2_info(ktx2[i], lens[i], &w, &h, &l); if (ri != UVOL_OK) { status[i] = ri; continue; } k = 0; }
    if (!have) { W0 = w; H0 = h; L0 = l; have = true; }
    else if (w != W0 || h != H0 || l != L0) continue;                       // another shape than the batch's: UVOL_E_INVALID
    if (k == 0 && target == UVOL_TARGET_ASTC) { status[i] = UVOL_E_UNSUPPORTED; continue; }      // (ASTC is the target of UASTC sources only; UASTC sources take every target)
    // an ETC1S batch is one launch per stage over files of ONE slice layout, so files with alpha slices (what this encoder's own
    // per-segment alpha re-run writes into an otherwise opaque sequence) are a batch of their own; the opaque targets refuse them in their slot
    if (k == 0 && tdec_file_alpha(ktx2[i], lens[i]) == 1) {
      if (target == UVOL_TARGET_ETC1 || target == UVOL_TARGET_BC1) { status[i] = UVOL_E_UNSUPPORTED; continue; }
      k = 2;
    }
    kind[i] = k; status[i] = UVOL_OK;
  }
  for (int k = 0; k < 3; k++) {
    std::vector<int> ix; for (int i = 0; i < n; i++) if (kind[i] == k) ix.push_back(i);
    if (ix.empty()) continue;
    std::vector<const uint8_t *> f(ix.size()); std::vector<size_t> ln(ix.size()); std::vector<uint8_t *> o(ix.size() * L0); std::vector<int> st(ix.size(), UVOL_OK);
    for (size_t j = 0; j < ix.size(); j++) { f[j] = ktx2[ix[j]]; ln[j] = lens[ix[j]]; for (uint32_t l = 0; l < L0; l++) o[j * L0 + l] = out[(size_t)ix[j] * L0 + l]; }
    const int rc = k == 1 ? tex_uastc_decode_segments(ctx, f.data(), ln.data(), (int)ix.size(), o.data(), layer_cap, outputs_on_device != 0, target, st.data())
                          : tex_decode_segments(ctx, f.data(), ln.data(), (int)ix.size(), o.data(), layer_cap, outputs_on_device != 0, target, st.data());
    if (rc == UVOL_E_HIP || rc == UVOL_E_NOSPACE) return rc;                 // the device / the caller's layer buffers: the call's failure
    for (size_t j = 0; j < ix.size(); j++) status[ix[j]] = rc != UVOL_OK ? rc : st[j];
  }
  return UVOL_OK;
}
int uvol_encode_texture_segments_st(uvol_ctx *ctx, const uint8_t *const *rgba, int n_segments, int n_layers, uint32_t width, uint32_t height,
                                    int inputs_on_device, uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !rgba || n_segments <= 0 || n_layers < 1 || !outs || !caps || !out_lens || !status) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  for (int s = 0; s < n_segments; s++) { status[s] = UVOL_OK; out_lens[s] = 0; }
  if (ctx->prm.uastc) return tex_uastc_encode_segments(ctx, rgba, n_segments, n_layers, width, height, inputs_on_device != 0, outs, caps, out_lens, status);
  return tex_encode_segments(ctx, rgba, n_segments, n_layers, width, height, inputs_on_device != 0, outs, caps, out_lens, status);
}
int uvol_decode_texture_segments(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments, uint8_t *const *rgba, size_t layer_cap) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !ktx2 || !lens || n_segments <= 0 || !rgba) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  return decode_dispatch(ctx, ktx2, lens, n_segments, rgba, layer_cap, false, 0);
}
int uvol_decode_texture_segments_dev(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments, uint8_t *const *rgba_dev, size_t layer_cap) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !ktx2 || !lens || n_segments <= 0 || !rgba_dev) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  return decode_dispatch(ctx, ktx2, lens, n_segments, rgba_dev, layer_cap, true, 0);
}
int uvol_transcode_texture_segments_etc1(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments, uint8_t *const *blocks, size_t layer_cap, int outputs_on_device) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !ktx2 || !lens || n_segments <= 0 || !blocks) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  return decode_dispatch(ctx, ktx2, lens, n_segments, blocks, layer_cap, outputs_on_device != 0, 1);
}

int uvol_transcode_texture_segments_bc7(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments, uint8_t *const *blocks, size_t layer_cap, int outputs_on_device) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !ktx2 || !lens || n_segments <= 0 || !blocks) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  return decode_dispatch(ctx, ktx2, lens, n_segments, blocks, layer_cap, outputs_on_device != 0, 2);
}

int uvol_transcode_texture_segments_etc2_rgba(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments, uint8_t *const *blocks, size_t layer_cap, int outputs_on_device) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !ktx2 || !lens || n_segments <= 0 || !blocks) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  return decode_dispatch(ctx, ktx2, lens, n_segments, blocks, layer_cap, outputs_on_device != 0, 4);
}

int uvol_transcode_texture_segments_astc(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments, uint8_t *const *blocks, size_t layer_cap, int outputs_on_device) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !ktx2 || !lens || n_segments <= 0 || !blocks) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  return decode_dispatch(ctx, ktx2, lens, n_segments, blocks, layer_cap, outputs_on_device != 0, 3);
}

int uvol_decode_mesh_batch(uvol_ctx *ctx, const uint8_t *const *drc, const size_t *lens, int n, uvol_decoded_mesh *out, int *status) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !drc || !lens || n < 0 || !out) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  const int mb = ctx->prm.max_batch;
  for (int b0 = 0; b0 < n; b0 += mb) {
    const int nb = n - b0 < mb ? n - b0 : mb;
    const int rc = geo_decode_batch(ctx, drc + b0, lens + b0, nb, out + b0, status ? status + b0 : nullptr);
    if (rc != UVOL_OK) return rc;
  }
  return UVOL_OK;
}

// (the profile entry points wait for enqueued work first: the worker thread updates the same records)
int uvol_decode_mesh_batch_dev(uvol_ctx *ctx, const uint8_t *const *drc, const size_t *lens, int n, uvol_decoded_mesh *out, int *status) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !drc || !lens || n < 0 || !out) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  const int mb = ctx->prm.max_batch;
  for (int b0 = 0; b0 < n; b0 += mb) {
    const int nb = n - b0 < mb ? n - b0 : mb;
    const int rc = geo_decode_batch(ctx, drc + b0, lens + b0, nb, out + b0, status ? status + b0 : nullptr, true);
    if (rc != UVOL_OK) return rc;
  }
  return UVOL_OK;
}

// (like every entry point it waits for this context's enqueued calls first: the worker thread and this call both borrow ctx->stream.  To
// parse batch b + 1 WHILE batch b encodes, parse on a second context of the same device - contexts are independent, device pointers are
// not tied to one -, as host/uvolenc.cpp does)
int uvol_parse_obj_batch_dev(uvol_ctx *ctx, const uint8_t *const *obj_text, const size_t *lens, int n, int slot, uvol_mesh *meshes_out, int *status) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !obj_text || !lens || n < 0 || !meshes_out) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  return obj_parse_batch(ctx, obj_text, lens, n, slot, meshes_out, status);
}

int uvol_unfilter_png_batch_dev(uvol_ctx *ctx, const uint8_t *const *inflated, int n, uint32_t width, uint32_t height, int channels, int slot, const uint8_t **rgba_dev_out) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !inflated || n < 0 || !rgba_dev_out) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  return png_unfilter_batch(ctx, inflated, n, width, height, channels, slot, rgba_dev_out);
}

int uvol_inflate_png_batch_dev(uvol_ctx *ctx, const uint8_t *const *zlib_streams, const size_t *lens, int n, uint32_t width, uint32_t height, int channels, int slot, const uint8_t **rgba_dev_out) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !zlib_streams || !lens || n < 0 || !rgba_dev_out) return UVOL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  return png_ingest_batch(ctx, zlib_streams, lens, n, width, height, channels, slot, rgba_dev_out);
}
int uvol_png_status(uvol_ctx *ctx, int slot, int *status, int n) {
  if (!ctx) return UVOL_E_INVALID;
  UVOL_AFTER_ASYNC(ctx);
  (void)hipSetDevice(ctx->device);
  return png_status(ctx, slot, status, n);
}

int uvol_encode_mesh_batch_dev_out(uvol_ctx *ctx, const uvol_mesh *meshes, int n, void *producer_stream,
                                   uint8_t *dev_out, size_t dev_cap, size_t *out_offs, size_t *out_lens, int *status) {
  UVOL_AFTER_ASYNC(ctx);
  if (!ctx || !meshes || n < 0 || !dev_out || !out_offs || !out_lens) return UVOL_E_INVALID;
  if (n > ctx->prm.max_batch) { ctx->set_error("uvol_encode_mesh_batch_dev_out: %d frames exceed max_batch %d (one packed output area per call)", n, ctx->prm.max_batch); return UVOL_E_INVALID; }
  (void)hipSetDevice(ctx->device);
  return geo_encode_batch_dev_out(ctx, meshes, n, (hipStream_t)producer_stream, dev_out, dev_cap, out_offs, out_lens, status);
}

int uvol_profile_enable(uvol_ctx *ctx, int on) { if (!ctx) return UVOL_E_INVALID; UVOL_AFTER_ASYNC(ctx); ctx->profiling = on != 0; return UVOL_OK; }
int uvol_profile_reset(uvol_ctx *ctx) {
  if (!ctx) return UVOL_E_INVALID;
  UVOL_AFTER_ASYNC(ctx);
  (void)hipStreamSynchronize(ctx->stream); ctx->resolve_profile(); ctx->prof.clear(); return UVOL_OK;
}
int uvol_profile_count(uvol_ctx *ctx) { if (!ctx) return 0; UVOL_AFTER_ASYNC(ctx); ctx->resolve_profile(); return (int)ctx->prof.size(); }
int uvol_profile_get(uvol_ctx *ctx, int i, char *name, size_t name_cap, uint64_t *launches, double *total_ms, uint64_t *algo_bytes) {
  if (ctx) UVOL_AFTER_ASYNC(ctx);
  if (!ctx || i < 0 || i >= (int)ctx->prof.size()) return UVOL_E_INVALID;
  const uvol_prof_entry &e = ctx->prof[i];
  if (name && name_cap) { strncpy(name, e.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (launches) *launches = e.launches;
  if (total_ms) *total_ms = e.total_ms;
  if (algo_bytes) *algo_bytes = e.algo_bytes;
  return UVOL_OK;
}

}  // extern "C"
