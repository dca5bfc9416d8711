// png_ingest.hip — PNG scanlines un-filtered on the device (the ingest stage).
//
// `basisu` reads the PNGs itself (scripts/Encoder.py:274-292).  On the host a 2048^2 RGBA PNG costs ~35 core-ms: the zlib inflate and
// the un-filter pass (Sub / Up / Average / Paeth recurrences over 16.8 MB).  The inflate stays on the host (a serial bit stream per file,
// and the files of a batch inflate in parallel on the ingest threads); the INFLATED scanlines - a filter-type byte + width * bpp filtered
// bytes per row - are uploaded as they are (the same 16.8 MB the un-filtered image would be) and un-filtered here, straight into the
// RGBA8 layers uvol_encode_texture_segments_dev reads.
//
// Average and Paeth are non-linear recurrences along a row AND depend on the row above, so a row cannot be split.  One WAVE per image:
// lane = channel * 16 + r, r = row of a band of 16 rows; row r runs ONE PIXEL behind row r - 1, so the pixel above (b) and above-left
// (c) are what the lane one below (same channel) produced one and two steps earlier - a DPP row shift, no LDS round trip -, the pixel to
// the left (a) is the lane's own last output.  A band takes width + 15 steps; the last row of a band stays in LDS for row 0 of the next.
// Every filter type runs the same straight-line step (the predictor is a select), so rows with different filters do not diverge.
// What a lone wave needs to run at ~0.1 us per step instead of 0.5 (tools/ingest_timing.py: it did not - 141 -> 134 ms per batch: the step is bound by instruction issue, ~1000 cycles for ~150 instructions of one wave): the filtered bytes
// of the next 16 steps are requested while the current 16 are processed (a global load per step was a memory round trip per step), and
// the outputs collect in an LDS ring (one byte write per lane and step) that the wave writes out 16 rows x 16 pixels at a time with
// 16-byte stores.
// Bit-identical to host/uvol_host.cpp read_png for 8-bit RGB / RGBA non-interlaced files (the other variants stay on the host).
#include "uvol_common.hpp"

struct PngJob { const uint8_t *raw; uint8_t *rgba; uint32_t w, h, ch; int32_t status; };

#define PNG_ROWS 16
__global__ void __launch_bounds__(64) k_png_unfilter(PngJob *jobs) {
  PngJob &J = jobs[blockIdx.x];
  UVOL_DYN_SMEM(uint32_t, lds);                            // [w] the band's last row (RGBA word per pixel), then [16 rows][32 pixels] output ring
  const uint32_t W = J.w, H = J.h, CH = J.ch; const size_t stride = (size_t)W * CH + 1;
  uint8_t *lastrow = reinterpret_cast<uint8_t *>(lds); uint8_t *ring = reinterpret_cast<uint8_t *>(lds + W);
  const int lane = (int)threadIdx.x, r = lane & 15, c = lane >> 4;
  const bool chan = (uint32_t)c < CH;                     // (RGB files: the alpha lane only supplies 255)
  for (uint32_t x = (uint32_t)lane; x < W; x += 64) lds[x] = 0;
  __syncthreads();
  const uint32_t T = 16 * ((W + 15) / 16 + 1);            // steps per band: every chunk of 16 pixels of every row has left the ring by then
  for (uint32_t band = 0; band < H; band += PNG_ROWS) {
    const uint32_t row = band + (uint32_t)r; const bool live = row < H;
    const uint8_t *src = J.raw + stride * (size_t)(live ? row : 0) + 1 + c;
    const int ft = live ? (int)src[-1 - c] : 0;
    // (a filter-type byte above 4 is not PNG; read_png leaves such a row as it is, and so does the select below)
    uint32_t o1 = 0, o2 = 0;                              // this lane's outputs one / two steps ago (pixels x - 1, x - 2 of its row)
    uint32_t cur[4] = { 0, 0, 0, 0 }, nxt[4];
    // filtered bytes of steps [s0, s0 + 16) of this lane (pixel = step - r), packed four per word
#define PNG_FETCH(dst, s0)                                                                                         \
    do {                                                                                                           \
      _Pragma("unroll") for (int w_ = 0; w_ < 4; w_++) {                                                            \
        uint32_t v_ = 0;                                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++) {                                                          \
          const int x_ = (int)(s0) + 4 * w_ + j_ - r;                                                              \
          const uint32_t b_ = (live && chan && x_ >= 0 && x_ < (int)W) ? (uint32_t)src[(size_t)x_ * CH] : 0u;     \
          v_ |= b_ << (8 * j_);                                                                                    \
        }                                                                                                          \
        dst[w_] = v_;                                                                                              \
      }                                                                                                            \
    } while (0)
    PNG_FETCH(nxt, 0);
    for (uint32_t t = 0; t < T; t++) {
      const uint32_t j = t & 15u;
      if (j == 0) { cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2]; cur[3] = nxt[3]; PNG_FETCH(nxt, t + 16); }
      const uint32_t wsel = j < 8 ? (j < 4 ? cur[0] : cur[1]) : (j < 12 ? cur[2] : cur[3]);
      const uint32_t f = (wsel >> (8 * (j & 3u))) & 255u;
      const int x = (int)t - r;                           // this row's pixel at this step
      const bool on = live && x >= 0 && x < (int)W;
      // above / above-left: the lane one below, one / two steps ago; row 0 of the band reads the previous band's last row from LDS
      uint32_t up = UVOL_ROW_SHR1(o1), ul = UVOL_ROW_SHR1(o2);
      if (r == 0) {
        up = (on && band) ? (uint32_t)lastrow[4 * x + c] : 0u;
        ul = (on && band && x > 0) ? (uint32_t)lastrow[4 * (x - 1) + c] : 0u;
      }
      if (x <= 0) ul = 0;
      const int ia = x > 0 ? (int)o1 : 0, ib = (int)up, ic = (int)ul;
      // predictor by filter type: 0 none, 1 a, 2 b, 3 (a + b) / 2, 4 Paeth(a, b, c)
      const int pa = ib - ic < 0 ? ic - ib : ib - ic, pb = ia - ic < 0 ? ic - ia : ia - ic, pcv = ia + ib - 2 * ic, pc = pcv < 0 ? -pcv : pcv;
      const int paeth = (pa <= pb && pa <= pc) ? ia : (pb <= pc ? ib : ic);
      const int pred = ft == 1 ? ia : (ft == 2 ? ib : (ft == 3 ? ((ia + ib) >> 1) : (ft == 4 ? paeth : 0)));
      uint32_t out = (f + (uint32_t)pred) & 255u;
      if (!chan) out = 255u;                              // alpha of an RGB file
      if (!on) out = 0;
      if (on) {
        ring[(((uint32_t)r * 32u + ((uint32_t)x & 31u)) << 2) + (uint32_t)c] = (uint8_t)out;
        if (r == PNG_ROWS - 1 || row == H - 1) lastrow[4 * x + c] = (uint8_t)out;
      }
      o2 = o1; o1 = out;
      if (j == 15u) {
        // 16 rows x 16 pixels leave the ring: lane = row * 4 + group of four pixels; the chunk of row rr that is complete now
        UVOL_WAVE_SYNC();
        const int rr = lane >> 2, qd = lane & 3;
        const int k = (int)((t + 1 - (uint32_t)rr) >> 4) - 1;
        const uint32_t orow = band + (uint32_t)rr, x0 = (uint32_t)(16 * k + 4 * qd);
        if (k >= 0 && orow < H && x0 < W) {
          const uint32_t *rp = lds + W + (uint32_t)rr * 32u + (x0 & 31u);
          uint32_t *dst = reinterpret_cast<uint32_t *>(J.rgba) + (size_t)orow * W + x0;
          if (x0 + 3 < W && (W & 3u) == 0) *reinterpret_cast<uint4 *>(dst) = make_uint4(rp[0], rp[1], rp[2], rp[3]);
          else for (uint32_t q = 0; q < 4 && x0 + q < W; q++) dst[q] = rp[q];
        }
        UVOL_WAVE_SYNC();
      }
    }
#undef PNG_FETCH
    __syncthreads();
  }
}

// ================================================================================================
// host side
// ================================================================================================
struct PngState { uvol_devbuf raw, jobs; uvol_devbuf rgba[2]; std::vector<PngJob> hjobs; hipStream_t stream = nullptr; hipEvent_t done[2] = { nullptr, nullptr }; bool pending[2] = { false, false }; };
int png_create(uvol_ctx *ctx) { ctx->png = new PngState(); return UVOL_OK; }
// the layers of the last un-filter call are ready before anything queued on `stream` from here on (the texture entry points call this
// before they read device inputs; uvol_sync too)
// (one event per slot: the encode of batch k waits for ITS slot, not for the un-filter of batch k + 1 queued meanwhile into the other one)
int png_order_before(uvol_ctx *ctx, hipStream_t stream, const uint8_t *const *layers, size_t n_layers) {
  PngState *S = ctx->png; if (!S) return UVOL_OK;
  // every layer of the call is looked at: a call may mix layers of both slots, or hold a pending slot's layer anywhere in its list
  bool need[2] = { false, false };
  for (int k = 0; k < 2; k++) {
    const uint8_t *b = (const uint8_t *)S->rgba[k].p;
    if (!S->pending[k] || !b) continue;
    for (size_t i = 0; i < n_layers && !need[k]; i++) need[k] = layers[i] >= b && layers[i] < b + S->rgba[k].cap;
  }
  for (int k = 0; k < 2; k++) if (need[k]) UVOL_HIP_CHECK(ctx, hipStreamWaitEvent(stream, S->done[k], 0));
  return UVOL_OK;
}
int png_wait(uvol_ctx *ctx) {
  PngState *S = ctx->png; int rc = UVOL_OK;
  if (S) for (int k = 0; k < 2; k++) if (S->pending[k]) {
    const hipError_t e = hipEventSynchronize(S->done[k]); S->pending[k] = false;
    if (e != hipSuccess && rc == UVOL_OK) { ctx->set_error("hipEventSynchronize (png un-filter slot %d): %s", k, hipGetErrorString(e)); rc = UVOL_E_HIP; }
  }
  return rc;
}
void png_destroy(uvol_ctx *ctx) {
  PngState *S = ctx->png; if (!S) return;
  if (S->stream) { (void)hipStreamSynchronize(S->stream); (void)hipStreamDestroy(S->stream); }
  for (hipEvent_t e : S->done) if (e) (void)hipEventDestroy(e);
  for (uvol_devbuf *b : { &S->raw, &S->jobs, &S->rgba[0], &S->rgba[1] }) if (b->p) (void)hipFree(b->p);
  delete S; ctx->png = nullptr;
}
// n images of one size: raw[i] = the INFLATED IDAT stream of an 8-bit non-interlaced RGB (channels 3) or RGBA (4) PNG, i.e. height rows
// of (1 filter-type byte + width * channels bytes), in host memory -> rgba_dev_out[i] = DEVICE pointer to width * height * 4 bytes,
// top row first, in the context's slot `slot` (valid until that slot is used again).  Runs on an ingest stream of its own and returns once
// the kernel is queued (the staged upload is what takes host time): the context's texture entry points order themselves behind it, so
// the un-filter of batch k + 1 runs beside the encode of batch k - one wave per image is 134 ms per batch whatever its size.
int png_unfilter_batch(uvol_ctx *ctx, const uint8_t *const *raw, int n, uint32_t w, uint32_t h, int channels, int slot, const uint8_t **rgba_dev_out) {
  PngState *S = ctx->png;
  if (n <= 0) return UVOL_OK;
  if (slot < 0 || slot > 1 || (channels != 3 && channels != 4) || !w || !h || w > 8192 || h > 16384) { ctx->set_error("uvol_unfilter_png_batch_dev: slot 0 / 1, 3 or 4 channels, at most 8192 x 16384 (wider images: un-filter on the host)"); return UVOL_E_INVALID; }
  const size_t rbytes = ((size_t)w * channels + 1) * h, ra = (rbytes + 4 + 255) & ~(size_t)255, obytes = (size_t)w * h * 4;
  int rc;
  if (!S->stream) { if (uvol_make_stream(ctx, &S->stream) != hipSuccess || hipEventCreateWithFlags(&S->done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&S->done[1], hipEventDisableTiming) != hipSuccess) { ctx->set_error("ingest stream: creation failed"); return UVOL_E_HIP; } }
  hipStream_t saved = ctx->stream; ctx->stream = S->stream;            // (uvol_ensure, uvol_upload_staged, Scope use ctx->stream)
  struct Restore { uvol_ctx *c; hipStream_t s; ~Restore() { c->stream = s; } } restore_{ ctx, saved };
  if ((rc = uvol_ensure(ctx, S->raw, ra * (size_t)n))) return rc;
  if ((rc = uvol_ensure(ctx, S->rgba[slot], obytes * (size_t)n))) return rc;
  if ((rc = uvol_ensure(ctx, S->jobs, sizeof(PngJob) * (size_t)n))) return rc;
  S->hjobs.assign((size_t)n, PngJob{});
  std::vector<UvolUpItem> ups; ups.reserve((size_t)n);
  for (int i = 0; i < n; i++) {
    if (!raw[i]) { ctx->set_error("PNG %d: no data", i); return UVOL_E_INVALID; }
    PngJob &J = S->hjobs[i]; J.raw = (const uint8_t *)S->raw.p + ra * (size_t)i; J.rgba = (uint8_t *)S->rgba[slot].p + obytes * (size_t)i; J.w = w; J.h = h; J.ch = (uint32_t)channels; J.status = 0;
    ups.push_back(UvolUpItem{ ra * (size_t)i, raw[i], rbytes });
    rgba_dev_out[i] = J.rgba;
  }
  { uvol_ctx::Scope sc(ctx, "ingest.png_upload", (uint64_t)rbytes * n);
    if ((rc = uvol_upload_staged(ctx, (uint8_t *)S->raw.p, ups))) return rc; }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(S->jobs.p, S->hjobs.data(), sizeof(PngJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  { uvol_ctx::Scope sc(ctx, "ingest.png_unfilter", (uint64_t)(rbytes + obytes) * n);
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch k_png_unfilter\n"); fflush(stderr); }
    hipLaunchKernelGGL(k_png_unfilter, dim3((unsigned)n), dim3(64), ((size_t)w + 16 * 32) * 4, ctx->stream, (PngJob *)S->jobs.p); }
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  UVOL_HIP_CHECK(ctx, hipEventRecord(S->done[slot], ctx->stream));
  S->pending[slot] = true;
  if (uvol_debug()) UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return UVOL_OK;
}
