// geo_scan.hpp - block scans and the scan kernels shared by the geometry stages.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
// ------------------------------------------------------------------------------------------------
// block-level exclusive scan (wave shuffles + LDS), blockDim.x == UVOL_BLOCK
// ------------------------------------------------------------------------------------------------
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *total) {
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

// sum of v over the workgroup; every thread of the block calls it (no early returns before it)
__device__ inline uint32_t block_sum(uint32_t v) {
  __shared__ uint32_t acc;
  if (threadIdx.x == 0) acc = 0;
  __syncthreads();
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(&acc, v);
  __syncthreads();
  const uint32_t r = acc;
  __syncthreads();
  return r;
}

// scan selectors.  The producers of the KEEP / ELIG / EVENTS flags write the per-block sums themselves (block_sum), so only
// SCAN_ORI still runs k_scan_blocks; k_scan_sums turns the sums into block offsets for all four.
enum { SCAN_KEEP = 0, SCAN_ELIG = 1, SCAN_ORI = 2, SCAN_EVENTS = 3, SCAN_SEQ = 4 };      // SCAN_SEQ: per input corner (sequential connectivity)
// NOTE: written as value-returning selects on purpose.  The earlier form (out-references assigned in
// an if/else chain) was miscompiled by hipcc 7.2 -O3 for gfx950: the sel==2 arm left the pointer
// register undefined ("implicit-def $sgpr8_sgpr9" in the ISA) and the kernel faulted at address 0.
__device__ __forceinline__ const uint8_t *scan_flags(const GeoJob &J, int sel) { return sel == SCAN_SEQ ? J.sq_flag : (sel == SCAN_KEEP ? J.keep : (sel == SCAN_EVENTS ? J.evcnt : (sel == SCAN_ELIG ? J.sbpack : J.has_ori))); }
__device__ __forceinline__ uint32_t scan_count(const GeoJob &J, int sel) { return sel == SCAN_SEQ ? 3u * J.nf_in : (sel == SCAN_KEEP ? J.nf_in : (sel == SCAN_EVENTS ? J.nf : (sel == SCAN_ELIG ? J.nf : (J.has_uv ? J.ne_uv : 0u)))); }
__global__ void __launch_bounds__(UVOL_BLOCK) k_scan_blocks(GeoJob *jobs, int sel) {
  GeoJob &J = jobs[blockIdx.y];
  const uint8_t *flags = scan_flags(J, sel); const uint32_t n = scan_count(J, sel);
  if (blockIdx.x >= uvol_blocks_dev(n)) return;       // block-uniform exit
  uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x;
  uint32_t v = (J.status == 0 && i < n) ? flags[i] : 0, tot;
  block_excl_scan(v, &tot);
  if (threadIdx.x == 0) (sel == SCAN_EVENTS ? J.bsum2 : J.bsum)[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_scan_sums(GeoJob *jobs, int sel) {
  GeoJob &J = jobs[blockIdx.y];
  const uint32_t nn = scan_count(J, sel);
  const uint32_t nblocks = uvol_blocks_dev(nn);
  uint32_t *bsum = sel == SCAN_EVENTS ? J.bsum2 : J.bsum;     // the event scan runs on the auxiliary stream
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nblocks; b0 += UVOL_BLOCK) {
    uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < nblocks ? bsum[i] : 0, tot;
    uint32_t ex = block_excl_scan(v, &tot);
    uint32_t c = carry;
    if (i < nblocks) bsum[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[nblocks] = carry;
}

// Corners per thread in the per-corner gather kernels (k_edge_match, k_aseg_a/b): they are latency-bound at full occupancy, so a
// thread issues every level of its dependent loads for GEO_ILP corners (one block stride apart: coalesced) before using any.
#define GEO_ILP 4
