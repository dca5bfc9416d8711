// png_ingest.hip — PNG scanlines un-filtered on the device (the ingest stage).
//
// `basisu` reads the PNGs itself (scripts/Encoder.py:274-292).  On the host a 2048^2 RGBA PNG costs ~35 core-ms: the zlib inflate and
// the un-filter pass (Sub / Up / Average / Paeth recurrences over 16.8 MB).  The inflate stays on the host (a serial bit stream per file,
// and the files of a batch inflate in parallel on the ingest threads); the INFLATED scanlines - a filter-type byte + width * bpp filtered
// bytes per row - are uploaded as they are (the same 16.8 MB the un-filtered image would be) and un-filtered here, straight into the
// RGBA8 layers uvol_encode_texture_segments_dev reads.
//
// Average and Paeth are non-linear recurrences along a row AND depend on the row above, so a row cannot be split.  One WAVE per image:
// lane (r, c) = row r of a band of 16 rows, channel c; row r runs ONE PIXEL behind row r - 1, so the pixel above (b) and above-left (c)
// are what lane (r - 1, c) produced one and two steps earlier (two wave shuffles), the pixel to the left (a) is the lane's own last
// output.  A band takes width + 15 steps; the last row of a band is kept in LDS for row 0 of the next one.  Every filter type runs
// the same straight-line step (the predictor is a select), so bands whose rows use different filters do not diverge.
// Bit-identical to host/uvol_host.cpp read_png for 8-bit RGB / RGBA non-interlaced files (the other variants stay on the host).
#include "uvol_common.hpp"

struct PngJob { const uint8_t *raw; uint8_t *rgba; uint32_t w, h, ch; int32_t status; };

#define PNG_ROWS 16
__global__ void __launch_bounds__(64) k_png_unfilter(PngJob *jobs) {
  PngJob &J = jobs[blockIdx.x];
  UVOL_DYN_SMEM(uint32_t, lastrow);                       // [w] the band's last row (RGBA as one word per pixel), read by row 0 of the next band
  const uint32_t W = J.w, H = J.h, CH = J.ch; const size_t stride = (size_t)W * CH + 1;
  const int lane = (int)threadIdx.x, r = lane >> 2, c = lane & 3;
  const bool chan = (uint32_t)c < CH;                     // (RGB files: the alpha lane only supplies 255)
  for (uint32_t x = (uint32_t)lane; x < W; x += 64) lastrow[x] = 0;
  __syncthreads();
  for (uint32_t band = 0; band < H; band += PNG_ROWS) {
    const uint32_t row = band + (uint32_t)r; const bool live = row < H;
    const uint8_t *src = J.raw + stride * (size_t)(live ? row : 0);
    const int ft = live ? src[0] : 0;
    // (a filter-type byte above 4 is not PNG; read_png leaves such a row as it is, and so does the select below)
    uint32_t o1 = 0, o2 = 0;                              // this lane's outputs one / two steps ago (pixels x - 1, x - 2 of its row)
    const uint32_t steps = W + PNG_ROWS - 1;
    for (uint32_t t = 0; t < steps; t++) {
      const int x = (int)t - r;                           // this row's pixel at this step
      const bool on = live && x >= 0 && x < (int)W;
      // above / above-left: lane (r - 1, c) one / two steps ago; row 0 of the band reads the previous band's last row from LDS
      uint32_t up = (uint32_t)__shfl_up((int)o1, 4), ul = (uint32_t)__shfl_up((int)o2, 4);
      if (r == 0) {
        const uint32_t wu = (on && band) ? lastrow[x] : 0u, wl = (on && band && x > 0) ? lastrow[x - 1] : 0u;
        up = (wu >> (8 * c)) & 255u; ul = (wl >> (8 * c)) & 255u;
      }
      if (x <= 0) ul = 0;
      const uint32_t a = x > 0 ? o1 : 0u;
      uint32_t f = (on && chan) ? (uint32_t)src[1 + (size_t)x * CH + c] : 0u;
      // predictor by filter type: 0 none, 1 a, 2 b, 3 (a + b) / 2, 4 Paeth(a, b, c)
      const int ia = (int)a, ib = (int)up, ic = (int)ul;
      const int pa = ib - ic < 0 ? ic - ib : ib - ic, pb = ia - ic < 0 ? ic - ia : ia - ic, pcv = ia + ib - 2 * ic, pc = pcv < 0 ? -pcv : pcv;
      const int paeth = (pa <= pb && pa <= pc) ? ia : (pb <= pc ? ib : ic);
      const int pred = ft == 1 ? ia : (ft == 2 ? ib : (ft == 3 ? ((ia + ib) >> 1) : (ft == 4 ? paeth : 0)));
      uint32_t out = (f + (uint32_t)pred) & 255u;
      if (!chan) out = 255u;                              // alpha of an RGB file
      if (!on) out = 0;
      // the four channel lanes of a row store one RGBA word
      const uint32_t g1 = (uint32_t)__shfl_down((int)out, 1), g2 = (uint32_t)__shfl_down((int)out, 2), g3 = (uint32_t)__shfl_down((int)out, 3);
      if (on && c == 0) {
        const uint32_t px = out | (g1 << 8) | (g2 << 16) | (g3 << 24);
        reinterpret_cast<uint32_t *>(J.rgba)[(size_t)row * W + (uint32_t)x] = px;
        if (r == PNG_ROWS - 1 || row == H - 1) lastrow[x] = px;
      }
      o2 = o1; o1 = out;
    }
    __syncthreads();
  }
}

// ================================================================================================
// host side
// ================================================================================================
struct PngState { uvol_devbuf raw, jobs; uvol_devbuf rgba[2]; std::vector<PngJob> hjobs; };
int png_create(uvol_ctx *ctx) { ctx->png = new PngState(); return UVOL_OK; }
void png_destroy(uvol_ctx *ctx) {
  PngState *S = ctx->png; if (!S) return;
  for (uvol_devbuf *b : { &S->raw, &S->jobs, &S->rgba[0], &S->rgba[1] }) if (b->p) (void)hipFree(b->p);
  delete S; ctx->png = nullptr;
}
// n images of one size: raw[i] = the INFLATED IDAT stream of an 8-bit non-interlaced RGB (channels 3) or RGBA (4) PNG, i.e. height rows
// of (1 filter-type byte + width * channels bytes), in host memory -> rgba_dev_out[i] = DEVICE pointer to width * height * 4 bytes,
// top row first, in the context's slot `slot` (valid until that slot is used again).  Runs on the context's stream.
int png_unfilter_batch(uvol_ctx *ctx, const uint8_t *const *raw, int n, uint32_t w, uint32_t h, int channels, int slot, const uint8_t **rgba_dev_out) {
  PngState *S = ctx->png;
  if (n <= 0) return UVOL_OK;
  if (slot < 0 || slot > 1 || (channels != 3 && channels != 4) || !w || !h || w > 8192 || h > 16384) { ctx->set_error("uvol_unfilter_png_batch_dev: slot 0 / 1, 3 or 4 channels, at most 8192 x 16384 (wider images: un-filter on the host)"); return UVOL_E_INVALID; }
  const size_t rbytes = ((size_t)w * channels + 1) * h, ra = (rbytes + 4 + 255) & ~(size_t)255, obytes = (size_t)w * h * 4;
  int rc;
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
    hipLaunchKernelGGL(k_png_unfilter, dim3((unsigned)n), dim3(64), (size_t)w * 4, ctx->stream, (PngJob *)S->jobs.p); }
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  UVOL_HIP_CHECK(ctx, hipMemcpy(S->hjobs.data(), S->jobs.p, sizeof(PngJob) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) if (S->hjobs[i].status != 0) { ctx->set_error("PNG %d: a scanline with a filter type above 4", i); return UVOL_E_INVALID; }
  return UVOL_OK;
}
