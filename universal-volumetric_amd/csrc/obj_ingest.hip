// obj_ingest.hip — OBJ text -> the arrays of uvol_mesh, on the device (the ingest stage).
//
// `draco_encoder -i frame.obj` (scripts/Encoder.py:256-262) parses the OBJ text itself: v / vt / vn / f lines, polygons fanned,
// 1-based and negative (relative) indices.  On the host that parse costs ~20 core-ms per 100 k-vertex frame (20.7 MB of text, 0.8 M
// numbers), which under a 16-CPU quota is a third of what bounds `uvolenc` from files.  Here the raw text is uploaded and parsed by the
// GPU: two passes over the bytes with the same thread -> byte mapping.
//   pass 1 (k_obj_count)  every thread looks at 16 bytes; a byte that starts a line classifies the line (v / vt / vn / f) and, for f,
//                         counts its corners; per-workgroup sums of {v, vt, vn, triangles}
//   scan   (k_obj_scan)   exclusive scan of the workgroup sums per frame -> totals (read back: the arrays are allocated exactly)
//   pass 2 (k_obj_parse)  the same threads rank their lines (workgroup base + scan inside the workgroup) and parse them straight
//                         into the arrays: numbers with the host parser's algorithm (host/uvol_host.cpp fast_float: <= 19 digits
//                         into a 64-bit integer, ONE correctly rounded double multiplication / division by an exact power of ten,
//                         narrowing to float unless the double sits next to a float rounding boundary), faces as fans
// The result is bit-identical to the host's read_obj() - which is bit-identical to strtof - or the frame is handed back: a number
// the fast path cannot decide (more digits, huge exponents, inf / nan, a value next to a rounding boundary, a subnormal), or a `v`
// line without three numbers, sets the frame's status to UVOL_E_UNSUPPORTED and the caller parses that file on the host.
#include "uvol_common.hpp"

#define OBJ_BPT 16                         // bytes per thread
#define OBJ_TILE (UVOL_BLOCK * OBJ_BPT)    // bytes per workgroup
enum { OBJ_V = 0, OBJ_VT = 1, OBJ_VN = 2, OBJ_TRI = 3 };
#define OBJ_E_HARD (-60)                   // irregular text: the host parser decides
#define OBJ_E_BADFACE (-61)                // a face references a missing vertex
#define OBJ_E_EMPTY (-62)                  // no faces / no positions

struct ObjJob {
  const uint8_t *text; uint32_t len, nblk;
  uint32_t *bcnt;                          // [4][nblk + 1] per-workgroup counts, then their exclusive scans (+ totals)
  uint32_t tot[4];
  float *pos, *uv, *nrm; uint32_t *ipos, *iuv, *inrm;
  int32_t status; uint32_t no_uv, no_n;
};

__device__ __forceinline__ bool obj_sp(uint8_t c) { return c == ' ' || c == '\t' || c == '\r'; }
// line kind at text[p..le): 0 v, 1 vt, 2 vn, 3 f, -1 other; q = first byte after the keyword
__device__ __forceinline__ int obj_kind(const uint8_t *t, uint32_t p, uint32_t le, uint32_t &q) {
  while (p < le && obj_sp(t[p])) p++;
  if (p + 1 < le && t[p] == 'v' && (t[p + 1] == ' ' || t[p + 1] == '\t')) { q = p + 1; return OBJ_V; }
  if (p + 2 < le && t[p] == 'v' && t[p + 1] == 't' && (t[p + 2] == ' ' || t[p + 2] == '\t')) { q = p + 2; return OBJ_VT; }
  if (p + 2 < le && t[p] == 'v' && t[p + 1] == 'n' && (t[p + 2] == ' ' || t[p + 2] == '\t')) { q = p + 2; return OBJ_VN; }
  if (p + 1 < le && t[p] == 'f' && (t[p + 1] == ' ' || t[p + 1] == '\t')) { q = p + 1; return 3; }
  return -1;
}
__device__ __forceinline__ uint32_t obj_line_end(const uint8_t *t, uint32_t p, uint32_t len) { while (p < len && t[p] != '\n') p++; return p; }
// optional sign + digits (what the host's integer() accepts)
__device__ __forceinline__ bool obj_int(const uint8_t *t, uint32_t &q, uint32_t le, long long &o) {
  const uint32_t s0 = q; bool neg = false;
  if (q < le && (t[q] == '-' || t[q] == '+')) { neg = t[q] == '-'; q++; }
  const uint32_t d0 = q; long long v = 0;
  while (q < le && t[q] >= '0' && t[q] <= '9') { if (v < (1ll << 40)) v = v * 10 + (t[q] - '0'); q++; }
  if (q == d0) { q = s0; return false; }
  o = neg ? -v : v; return true;
}
// corners of an f line (the loop of read_obj: skip blanks, an integer, optional /b/c)
__device__ inline uint32_t obj_face_corners(const uint8_t *t, uint32_t q, uint32_t le) {
  uint32_t k = 0;
  for (;;) {
    while (q < le && obj_sp(t[q])) q++;
    if (q >= le) break;
    long long a; if (!obj_int(t, q, le, a)) break;
    if (q < le && t[q] == '/') { q++; long long b; if (q < le && t[q] != '/') (void)obj_int(t, q, le, b); if (q < le && t[q] == '/') { q++; (void)obj_int(t, q, le, b); } }
    k++;
  }
  return k;
}
__device__ const double OBJ_P10[23] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22 };
// one number at t[q..le): 1 parsed (q advanced), 0 nothing there, -1 the host must decide (host/uvol_host.cpp: fast_float + its strtof fallback)
__device__ inline int obj_num(const uint8_t *t, uint32_t &q, uint32_t le, float &out) {
  while (q < le && obj_sp(t[q])) q++;
  if (q >= le) return 0;
  uint32_t p = q; bool neg = false;
  if (t[p] == '-' || t[p] == '+') { neg = t[p] == '-'; p++; }
  unsigned long long mant = 0; int nd = 0, e10 = 0; bool any = false;
  while (p < le && t[p] >= '0' && t[p] <= '9') { if (nd < 19) { mant = mant * 10 + (unsigned long long)(t[p] - '0'); if (mant) nd++; } else e10++; any = true; p++; }
  if (p < le && t[p] == '.') { p++; while (p < le && t[p] >= '0' && t[p] <= '9') { if (nd < 19) { mant = mant * 10 + (unsigned long long)(t[p] - '0'); if (mant) nd++; e10--; } any = true; p++; } }
  if (!any) return -1;
  const bool exact = nd < 19;
  if (p < le && (t[p] == 'e' || t[p] == 'E')) {
    uint32_t pe = p + 1; bool en = false; if (pe < le && (t[pe] == '-' || t[pe] == '+')) { en = t[pe] == '-'; pe++; }
    if (pe < le && t[pe] >= '0' && t[pe] <= '9') { int ev = 0; while (pe < le && t[pe] >= '0' && t[pe] <= '9') { if (ev < 10000) ev = ev * 10 + (t[pe] - '0'); pe++; } e10 += en ? -ev : ev; p = pe; }
  }
  if (p < le && ((t[p] >= 'a' && t[p] <= 'z') || (t[p] >= 'A' && t[p] <= 'Z'))) return -1;
  if (!exact || mant > (1ull << 53) || e10 < -22 || e10 > 22) return -1;
  double d = (double)mant; d = e10 < 0 ? d / OBJ_P10[-e10] : d * OBJ_P10[e10];
  unsigned long long bits; memcpy(&bits, &d, 8);
  const uint32_t low = (uint32_t)(bits & 0x1fffffffu);
  if (low - 0x0ffffffeu <= 4u) return -1;                                  // next to a float rounding boundary
  const float f = (float)d;
  if (!(fabsf(f) >= 1.17549435e-38f) && mant != 0) return -1;              // subnormal
  out = neg ? -f : f; q = p; return 1;
}

// The workgroup's tile (+ the 16 bytes before it and OBJ_OVER bytes behind it) staged in LDS with 16-byte loads: the parsers walk their
// lines byte by byte, and a byte load from global memory is a cache round trip per byte and lane (k_obj_parse read the text at 16 GB/s;
// VERDICT r4 item 6c).  The returned pointer is the tile's first byte: it is indexed with positions RELATIVE to the tile, from -1 (the
// byte before the tile) to min(len, tile0 + OBJ_TILE + OBJ_OVER) - tile0; a line that runs past that window is read from global memory
// (absolute positions) as before.
#define OBJ_OVER 496
#define OBJ_STAGE (16 + OBJ_TILE + OBJ_OVER)
__device__ __forceinline__ const uint8_t *obj_stage(const uint8_t *text, uint32_t len, uint32_t tile0, uint8_t *stage) {
  for (uint32_t j = threadIdx.x; j < OBJ_STAGE / 16; j += UVOL_BLOCK) {
    const long long gp = (long long)tile0 - 16 + 16ll * j;
    if (gp >= 0 && gp < (long long)len) *reinterpret_cast<uint4 *>(stage + 16 * j) = *reinterpret_cast<const uint4 *>(text + gp);
  }
  __syncthreads();
  return stage + 16;
}
// the line that starts at file position i: t / li = where to read it (the staged tile with a relative position, or - a long line - the
// file with the absolute one); returns its end in the same coordinates
__device__ __forceinline__ uint32_t obj_line_window(const uint8_t *ts, const uint8_t *tg, uint32_t i, uint32_t tile0, uint32_t wend, uint32_t len, const uint8_t *&t, uint32_t &li) {
  const uint32_t le = obj_line_end(ts, i - tile0, wend - tile0);
  if (le == wend - tile0 && wend < len) { t = tg; li = i; return obj_line_end(tg, wend, len); }
  t = ts; li = i - tile0; return le;
}

// pass 1: per-workgroup counts
__global__ void __launch_bounds__(UVOL_BLOCK) k_obj_count(ObjJob *jobs) {
  ObjJob &J = jobs[blockIdx.y];
  if (blockIdx.x >= J.nblk) return;
  const uint8_t *tg = J.text; const uint32_t len = J.len;
  __shared__ __attribute__((aligned(16))) uint8_t stage[OBJ_STAGE];
  const uint32_t tile0 = blockIdx.x * OBJ_TILE, wend = len - tile0 > OBJ_TILE + OBJ_OVER ? tile0 + OBJ_TILE + OBJ_OVER : len;
  const uint8_t *ts = obj_stage(tg, len, tile0, stage);
  const uint32_t b0 = tile0 + threadIdx.x * OBJ_BPT;
  uint32_t c[4] = { 0, 0, 0, 0 };
  for (uint32_t i = b0; i < b0 + OBJ_BPT && i < len; i++) {
    if (i != 0 && ts[(int)(i - tile0) - 1] != '\n') continue;
    const uint8_t *t; uint32_t li; const uint32_t le = obj_line_window(ts, tg, i, tile0, wend, len, t, li); uint32_t q;
    const int k = obj_kind(t, li, le, q);
    if (k == 3) { const uint32_t nc = obj_face_corners(t, q, le); c[OBJ_TRI] += nc > 2 ? nc - 2 : 0; }
    else if (k >= 0) c[k]++;
  }
  __shared__ uint32_t acc[4];
  if (threadIdx.x < 4) acc[threadIdx.x] = 0;
  __syncthreads();
  for (int k = 0; k < 4; k++) { uint32_t v = c[k]; for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d); if ((threadIdx.x & 63) == 0 && v) atomicAdd(&acc[k], v); }
  __syncthreads();
  if (threadIdx.x < 4) J.bcnt[(size_t)threadIdx.x * (J.nblk + 1) + blockIdx.x] = acc[threadIdx.x];
}
// exclusive scan of the workgroup counts (one workgroup per frame and kind)
__global__ void __launch_bounds__(UVOL_BLOCK) k_obj_scan(ObjJob *jobs) {
  ObjJob &J = jobs[blockIdx.y];
  uint32_t *cnt = J.bcnt + (size_t)blockIdx.x * (J.nblk + 1); const uint32_t m = J.nblk;
  __shared__ uint32_t wsum[UVOL_BLOCK / 64]; __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < m; b0 += UVOL_BLOCK) {
    const uint32_t i = b0 + threadIdx.x; const uint32_t v = i < m ? cnt[i] : 0;
    uint32_t x = v; const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0; for (int k = 0; k < UVOL_BLOCK / 64; k++) { if (k < w) base += wsum[k]; tot += wsum[k]; }
    const uint32_t c = carry;
    if (i < m) cnt[i] = c + base + x - v;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) { cnt[m] = carry; J.tot[blockIdx.x] = carry; }
}
// pass 2: rank and parse
__global__ void __launch_bounds__(UVOL_BLOCK) k_obj_parse(ObjJob *jobs) {
  ObjJob &J = jobs[blockIdx.y];
  if (blockIdx.x >= J.nblk) return;
  const uint8_t *tg = J.text; const uint32_t len = J.len;
  __shared__ __attribute__((aligned(16))) uint8_t stage[OBJ_STAGE];
  const uint32_t tile0 = blockIdx.x * OBJ_TILE, wend = len - tile0 > OBJ_TILE + OBJ_OVER ? tile0 + OBJ_TILE + OBJ_OVER : len;
  const uint8_t *ts = obj_stage(tg, len, tile0, stage);
  const uint32_t b0 = tile0 + threadIdx.x * OBJ_BPT;
  const bool live = J.status == 0;
  // this thread's lines: counts first (as in pass 1), then an exclusive scan over the workgroup gives the rank of its first line of each kind
  uint32_t c[4] = { 0, 0, 0, 0 };
  if (live) for (uint32_t i = b0; i < b0 + OBJ_BPT && i < len; i++) {
    if (i != 0 && ts[(int)(i - tile0) - 1] != '\n') continue;
    const uint8_t *t; uint32_t li; const uint32_t le = obj_line_window(ts, tg, i, tile0, wend, len, t, li); uint32_t q;
    const int k = obj_kind(t, li, le, q);
    if (k == 3) { const uint32_t nc = obj_face_corners(t, q, le); c[OBJ_TRI] += nc > 2 ? nc - 2 : 0; }
    else if (k >= 0) c[k]++;
  }
  __shared__ uint32_t wsum[4][UVOL_BLOCK / 64];
  uint32_t rank[4];
  { const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x[4];
    for (int k = 0; k < 4; k++) { x[k] = c[k]; for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x[k], d); if (lane >= d) x[k] += y; } if (lane == 63) wsum[k][w] = x[k]; }
    __syncthreads();
    for (int k = 0; k < 4; k++) { uint32_t base = 0; for (int j = 0; j < w; j++) base += wsum[k][j]; rank[k] = J.bcnt[(size_t)k * (J.nblk + 1) + blockIdx.x] + base + x[k] - c[k]; } }
  if (!live) return;
  const uint32_t NP = J.tot[OBJ_V], NT = J.tot[OBJ_VT], NN = J.tot[OBJ_VN];
  for (uint32_t i = b0; i < b0 + OBJ_BPT && i < len; i++) {
    if (i != 0 && ts[(int)(i - tile0) - 1] != '\n') continue;
    const uint8_t *t; uint32_t li; const uint32_t le = obj_line_window(ts, tg, i, tile0, wend, len, t, li); uint32_t q;
    const int k = obj_kind(t, li, le, q);
    if (k == OBJ_V) {
      float v[3]; bool ok = true;
      for (int j = 0; j < 3; j++) { const int r = obj_num(t, q, le, v[j]); if (r != 1) { ok = false; break; } }
      if (!ok) { J.status = OBJ_E_HARD; rank[OBJ_V]++; continue; }        // (a `v` line without three plain numbers: the host decides whether it counts)
      float *o = J.pos + 3 * (size_t)rank[OBJ_V]; o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; rank[OBJ_V]++;
    } else if (k == OBJ_VT) {
      float v[2] = { 0.f, 0.f };
      for (int j = 0; j < 2; j++) { const int r = obj_num(t, q, le, v[j]); if (r < 0) { J.status = OBJ_E_HARD; break; } if (r == 0) break; }
      float *o = J.uv + 2 * (size_t)rank[OBJ_VT]; o[0] = v[0]; o[1] = v[1]; rank[OBJ_VT]++;
    } else if (k == OBJ_VN) {
      float v[3] = { 0.f, 0.f, 0.f };
      for (int j = 0; j < 3; j++) { const int r = obj_num(t, q, le, v[j]); if (r < 0) { J.status = OBJ_E_HARD; break; } if (r == 0) break; }
      float *o = J.nrm + 3 * (size_t)rank[OBJ_VN]; o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; rank[OBJ_VN]++;
    } else if (k == 3) {
      // indices relative to what has been defined SO FAR (the ranks of this line), fans over the corners
      const long long np = rank[OBJ_V], nt = rank[OBJ_VT], nn = rank[OBJ_VN];
      long long f0[3] = { 0, 0, 0 }, pr[3] = { 0, 0, 0 }; uint32_t nc = 0;
      for (;;) {
        while (q < le && obj_sp(t[q])) q++;
        if (q >= le) break;
        long long a; if (!obj_int(t, q, le, a)) break;
        long long b = 0, cc = 0; bool hb = false, hc = false;
        if (q < le && t[q] == '/') { q++; if (q < le && t[q] != '/') hb = obj_int(t, q, le, b); if (q < le && t[q] == '/') { q++; hc = obj_int(t, q, le, cc); } }
        const long long cur[3] = { a < 0 ? np + a : a - 1, hb ? (b < 0 ? nt + b : b - 1) : -1, hc ? (cc < 0 ? nn + cc : cc - 1) : -1 };
        if (nc == 0) { f0[0] = cur[0]; f0[1] = cur[1]; f0[2] = cur[2]; }
        else if (nc >= 2) {
          const size_t o = 3 * (size_t)rank[OBJ_TRI];
          const long long *tri[3] = { f0, pr, cur };
          for (int j = 0; j < 3; j++) {
            const long long *x = tri[j];
            if (x[0] < 0 || x[0] >= np) { J.status = OBJ_E_BADFACE; }
            J.ipos[o + j] = (uint32_t)(x[0] < 0 ? 0 : x[0]);
            if (x[1] < 0 || x[1] >= nt) J.no_uv = 1;
            J.iuv[o + j] = x[1] < 0 ? 0u : (uint32_t)x[1];
            if (x[2] < 0 || x[2] >= nn) J.no_n = 1;
            J.inrm[o + j] = x[2] < 0 ? 0u : (uint32_t)x[2];
          }
          rank[OBJ_TRI]++;
        }
        pr[0] = cur[0]; pr[1] = cur[1]; pr[2] = cur[2]; nc++;
      }
    }
  }
  (void)NP; (void)NT; (void)NN;
}

// ================================================================================================
// host side
// ================================================================================================
struct ObjState {
  hipStream_t stream = nullptr;
  uvol_devbuf text, jobs, cnt;
  uvol_devbuf arrays[2];                     // two slots: the arrays of one batch stay valid while the next batch is parsed
  std::vector<ObjJob> hjobs;
};
int obj_create(uvol_ctx *ctx) { ctx->obj = new ObjState(); return UVOL_OK; }
void obj_destroy(uvol_ctx *ctx) {
  ObjState *S = ctx->obj; if (!S) return;
  if (S->stream) { (void)hipStreamSynchronize(S->stream); (void)hipStreamDestroy(S->stream); }
  for (uvol_devbuf *b : { &S->text, &S->jobs, &S->cnt, &S->arrays[0], &S->arrays[1] }) if (b->p) (void)hipFree(b->p);
  delete S; ctx->obj = nullptr;
}

#define OLAUNCH(k, grid, block, ...)                                                             \
  do {                                                                                           \
    if (uvol_debug()) { fprintf(stderr, "[uvol] launch %s\n", #k); fflush(stderr); }              \
    hipLaunchKernelGGL(k, grid, block, 0, ctx->stream, __VA_ARGS__);                             \
    if (uvol_debug()) { hipError_t e_ = hipStreamSynchronize(ctx->stream); if (e_ != hipSuccess) { fprintf(stderr, "[uvol] %s FAILED: %s\n", #k, hipGetErrorString(e_)); fflush(stderr); } } \
  } while (0)

// n OBJ files as text (host memory) -> meshes_out[i] with DEVICE pointers into the context's slot `slot` (valid until that slot is parsed
// into again); status[i]: UVOL_OK, UVOL_E_UNSUPPORTED (irregular text: parse it on the host), UVOL_E_INVALID (a face references a missing
// vertex / no faces, as read_obj reports)
int obj_parse_batch(uvol_ctx *ctx, const uint8_t *const *texts, const size_t *lens, int n, int slot, uvol_mesh *meshes_out, int *status) {
  ObjState *S = ctx->obj;
  if (n <= 0) return UVOL_OK;
  if (slot < 0 || slot > 1) { ctx->set_error("uvol_parse_obj_batch_dev: slot must be 0 or 1"); return UVOL_E_INVALID; }
  if (!S->stream && uvol_make_stream(ctx, &S->stream) != hipSuccess) { ctx->set_error("ingest stream: creation failed"); return UVOL_E_HIP; }
  hipStream_t saved = ctx->stream; ctx->stream = S->stream;            // (uvol_ensure, uvol_upload_staged, Scope use ctx->stream)
  struct Restore { uvol_ctx *c; hipStream_t s; ~Restore() { c->stream = s; } } restore_{ ctx, saved };
  S->hjobs.assign((size_t)n, ObjJob{});
  std::vector<size_t> toff((size_t)n), coff((size_t)n);
  size_t ttot = 0, ctot = 0; uint32_t max_blk = 0;
  for (int i = 0; i < n; i++) {
    if (!texts[i] || lens[i] == 0 || lens[i] > 0xfffffff0ull) { ctx->set_error("OBJ text %d: empty or larger than 4 GB", i); return UVOL_E_INVALID; }
    ObjJob &J = S->hjobs[i]; J.len = (uint32_t)lens[i]; J.nblk = (uint32_t)((lens[i] + OBJ_TILE - 1) / OBJ_TILE);
    toff[i] = ttot; ttot += (lens[i] + 255) & ~(size_t)255;
    coff[i] = ctot; ctot += 4 * ((size_t)J.nblk + 1) * sizeof(uint32_t);
    max_blk = std::max(max_blk, J.nblk);
  }
  int rc;
  if ((rc = uvol_ensure(ctx, S->text, ttot + 256))) return rc;
  if ((rc = uvol_ensure(ctx, S->cnt, ctot))) return rc;
  if ((rc = uvol_ensure(ctx, S->jobs, sizeof(ObjJob) * (size_t)n))) return rc;
  std::vector<UvolUpItem> ups; ups.reserve((size_t)n);
  for (int i = 0; i < n; i++) { ObjJob &J = S->hjobs[i]; J.text = (const uint8_t *)S->text.p + toff[i]; J.bcnt = (uint32_t *)((uint8_t *)S->cnt.p + coff[i]); ups.push_back(UvolUpItem{ toff[i], texts[i], lens[i] }); }
  { uvol_ctx::Scope sc(ctx, "ingest.obj_upload", (uint64_t)ttot);
    if ((rc = uvol_upload_staged(ctx, (uint8_t *)S->text.p, ups))) return rc; }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(S->jobs.p, S->hjobs.data(), sizeof(ObjJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ObjJob *dj = (ObjJob *)S->jobs.p;
  { uvol_ctx::Scope sc(ctx, "ingest.obj_count", (uint64_t)ttot);
    OLAUNCH(k_obj_count, dim3(max_blk, (unsigned)n), dim3(UVOL_BLOCK), dj);
    OLAUNCH(k_obj_scan, dim3(4, (unsigned)n), dim3(UVOL_BLOCK), dj); }
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  UVOL_HIP_CHECK(ctx, hipMemcpy(S->hjobs.data(), dj, sizeof(ObjJob) * (size_t)n, hipMemcpyDeviceToHost));
  // the arrays, exactly sized
  auto a256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  std::vector<size_t> aoff((size_t)n); size_t atot = 0;
  for (int i = 0; i < n; i++) {
    const ObjJob &J = S->hjobs[i];
    aoff[i] = atot;
    atot += a256(12 * (size_t)J.tot[OBJ_V] + 16) + a256(8 * (size_t)J.tot[OBJ_VT] + 16) + a256(12 * (size_t)J.tot[OBJ_VN] + 16) + 3 * a256(12 * (size_t)J.tot[OBJ_TRI] + 16);
  }
  if ((rc = uvol_ensure(ctx, S->arrays[slot], atot))) return rc;
  for (int i = 0; i < n; i++) {
    ObjJob &J = S->hjobs[i]; uint8_t *b = (uint8_t *)S->arrays[slot].p + aoff[i];
    J.pos = (float *)b; b += a256(12 * (size_t)J.tot[OBJ_V] + 16);
    J.uv = (float *)b; b += a256(8 * (size_t)J.tot[OBJ_VT] + 16);
    J.nrm = (float *)b; b += a256(12 * (size_t)J.tot[OBJ_VN] + 16);
    J.ipos = (uint32_t *)b; b += a256(12 * (size_t)J.tot[OBJ_TRI] + 16);
    J.iuv = (uint32_t *)b; b += a256(12 * (size_t)J.tot[OBJ_TRI] + 16);
    J.inrm = (uint32_t *)b;
    J.status = 0; J.no_uv = 0; J.no_n = 0;
  }
  UVOL_HIP_CHECK(ctx, hipMemcpyAsync(dj, S->hjobs.data(), sizeof(ObjJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  { uvol_ctx::Scope sc(ctx, "ingest.obj_parse", (uint64_t)ttot);
    OLAUNCH(k_obj_parse, dim3(max_blk, (unsigned)n), dim3(UVOL_BLOCK), dj); }
  UVOL_HIP_CHECK(ctx, hipGetLastError());
  UVOL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  UVOL_HIP_CHECK(ctx, hipMemcpy(S->hjobs.data(), dj, sizeof(ObjJob) * (size_t)n, hipMemcpyDeviceToHost));
  ctx->resolve_profile();
  int worst = UVOL_OK;
  for (int i = 0; i < n; i++) {
    const ObjJob &J = S->hjobs[i]; uvol_mesh &M = meshes_out[i]; memset(&M, 0, sizeof M);
    int st = UVOL_OK;
    if (J.status == OBJ_E_HARD) st = UVOL_E_UNSUPPORTED;
    else if (J.status != 0 || J.tot[OBJ_TRI] == 0 || J.tot[OBJ_V] == 0) st = UVOL_E_INVALID;
    if (st == UVOL_OK) {
      M.pos = J.pos; M.n_pos = J.tot[OBJ_V]; M.idx_pos = J.ipos; M.n_faces = J.tot[OBJ_TRI];
      if (!J.no_uv && J.tot[OBJ_VT]) { M.uv = J.uv; M.n_uv = J.tot[OBJ_VT]; M.idx_uv = J.iuv; }
      if (!J.no_n && J.tot[OBJ_VN]) { M.nrm = J.nrm; M.n_nrm = J.tot[OBJ_VN]; M.idx_nrm = J.inrm; }
    } else if (st == UVOL_E_UNSUPPORTED) ctx->set_error("OBJ text %d: a number or line the device parser leaves to the host (more than 19 digits, inf / nan, a value on a float rounding boundary, an incomplete `v` line)", i);
    else ctx->set_error("OBJ text %d: %s", i, J.status == OBJ_E_BADFACE ? "face references a missing vertex" : "no faces");
    if (status) status[i] = st;
    if (st != UVOL_OK) worst = st;
  }
  return status ? UVOL_OK : worst;
}
