// latbench.hip — pointer-chase microbenchmark for the serial walkers' access pattern (diagnostic tool, not product).
// Each one-lane workgroup chases a random cycle through its own region of 32-byte records.
// Variants: 0 plain chase; 1 + one extra independent random 32 B load per step (speculative prefetch cost);
//           2 + one sequential 4 B store per step; 3 + store + ~40 dependent ALU ops; 4 records in one shared region.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
typedef int v4i __attribute__((ext_vector_type(4)));
#define G(T) __attribute__((address_space(1))) T *
__global__ void __launch_bounds__(64) chase(const int4 *recs, int *out, long long *clk, int nrec, int steps, int variant, int shared_region) {
  if (variant == 8) {       // two independent chains on two LANES of the same wave (one vector load instruction per step)
    if (threadIdx.x > 1) return;
    G(const v4i) r8 = (G(const v4i))(recs + 2 * (size_t)blockIdx.x * nrec);
    int idx8 = (int)((blockIdx.x * 7919u + threadIdx.x * 104729u) % (unsigned)nrec);
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; s++) { const v4i a = r8[2 * (size_t)idx8]; idx8 = a.x; }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    out[(size_t)blockIdx.x * steps + threadIdx.x] = idx8;
    return;
  }
  if (variant == 11 || variant == 12) {
    // 11: chase + ~60 dependent scalar-ish ALU ops per step on the loaded value (a stand-in for the walkers' bookkeeping)
    // 12: same work, but the NEXT step's record is requested as soon as its index is known (start of the step) through a
    //     formally divergent address, and only moved to scalars (readfirstlane) when the following step needs it
    if (threadIdx.x) return;
    G(const v4i) rr = (G(const v4i))(recs + 2 * (size_t)blockIdx.x * nrec);
    const int dz = (int)__builtin_amdgcn_mbcnt_lo(~0u, 0u);
    int idx = (int)((blockIdx.x * 7919u) % (unsigned)nrec), acc = 0;
    const long long t0 = wall_clock64();
    if (variant == 11) {
      for (int s = 0; s < steps; s++) {
        const v4i a = rr[2 * (size_t)idx];
        int w = a.y; for (int k = 0; k < 60; k++) w = (w * 5 + k) ^ (w >> 3);
        acc += w; idx = a.x;
      }
    } else {
      v4i cur = rr[2 * (size_t)idx];                 // record of the current index
      for (int s = 0; s < steps; s++) {
        const int nxt = __builtin_amdgcn_readfirstlane(cur.x), payload = __builtin_amdgcn_readfirstlane(cur.y);
        const v4i pre = rr[2 * (size_t)(nxt + dz)];  // in flight during the bookkeeping below
        int w = payload; for (int k = 0; k < 60; k++) w = (w * 5 + k) ^ (w >> 3);
        acc += w; idx = nxt; cur = pre;
      }
    }
    const long long t1 = wall_clock64();
    clk[blockIdx.x] = t1 - t0; out[(size_t)blockIdx.x * steps] = idx; out[(size_t)blockIdx.x * steps + 1] = acc;
    return;
  }
  if (threadIdx.x) return;
  if (variant == 9 || variant == 10) {   // chase through LDS-DMA slots: 9 = load the needed record only; 10 = speculative pair issued early + 40 ALU ops
    extern __shared__ v4i lds[];
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) v4i *)lds;
    __attribute__((address_space(3))) v4i *L = (__attribute__((address_space(3))) v4i *)lds;
    G(const v4i) r9 = (G(const v4i))(recs + 2 * (size_t)blockIdx.x * nrec);
    int idx9 = (int)((blockIdx.x * 7919u) % (unsigned)nrec), acc9 = 0; unsigned tmp;
    const long long t0 = wall_clock64();
    if (variant == 9) {
      for (int s = 0; s < steps; s++) {
        const v4i *p = (const v4i *)(recs + 2 * ((size_t)blockIdx.x * nrec + idx9));
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, off\n\tglobal_load_lds_dwordx4 %2, off offset:16\n\ts_mov_b32 m0, %0" : "=&s"(tmp) : "s"(lbase), "v"(p) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const v4i a = L[0], b = L[1]; idx9 = a.x; acc9 += b.x;
      }
    } else {
      // slot 0/1 = record of the next index (the one needed), slot 2/3 = a second speculative record
      int have = -1;
      for (int s = 0; s < steps; s++) {
        if (have != idx9) {
          const v4i *p = (const v4i *)(recs + 2 * ((size_t)blockIdx.x * nrec + idx9));
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, off\n\tglobal_load_lds_dwordx4 %2, off offset:16\n\ts_mov_b32 m0, %0" : "=&s"(tmp) : "s"(lbase), "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const v4i a = L[0]; 
        // issue the speculative pair for the next step right away (a.x is the true successor, a.y a decoy)
        const v4i *p0 = (const v4i *)(recs + 2 * ((size_t)blockIdx.x * nrec + a.x)), *p1 = (const v4i *)(recs + 2 * ((size_t)blockIdx.x * nrec + a.y));
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, off\n\tglobal_load_lds_dwordx4 %2, off offset:16\n\ts_mov_b32 m0, %3\n\tglobal_load_lds_dwordx4 %4, off\n\tglobal_load_lds_dwordx4 %4, off offset:16\n\ts_mov_b32 m0, %0"
                     : "=&s"(tmp) : "s"(lbase), "v"(p0), "s"(lbase + 32), "v"(p1) : "memory");
        have = a.x;
        int nx = a.x; for (int k = 0; k < 40; k++) nx = (nx ^ (nx >> 3)) + k - ((nx + k) ^ ((nx + k) >> 3));
        acc9 += nx & 0; idx9 = a.x;
      }
    }
    const long long t1 = wall_clock64();
    clk[blockIdx.x] = t1 - t0; out[(size_t)blockIdx.x * steps] = idx9; out[(size_t)blockIdx.x * steps + 1] = acc9;
    return;
  }
  if (variant == 7) {       // two independent chains in ONE lane (two load instructions in flight)
    G(const v4i) r7 = (G(const v4i))(recs + 2 * (size_t)blockIdx.x * nrec);
    int i0 = (int)((blockIdx.x * 7919u) % (unsigned)nrec), i1 = (int)((blockIdx.x * 7919u + 104729u) % (unsigned)nrec);
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; s++) { const v4i a = r7[2 * (size_t)i0], b = r7[2 * (size_t)i1]; i0 = a.x; i1 = b.x; }
    const long long t1 = wall_clock64();
    clk[blockIdx.x] = t1 - t0; out[(size_t)blockIdx.x * steps] = i0 + i1;
    return;
  }
  G(const v4i) r = (G(const v4i))(recs + 2 * (size_t)(shared_region ? 0 : blockIdx.x) * nrec);
  G(int) o = (G(int))(out + (size_t)blockIdx.x * steps);
  int idx = (int)((blockIdx.x * 7919u) % (unsigned)nrec), acc = 0;
  v4i pb = r[1], pc = r[3];
  const long long t0 = wall_clock64();
  for (int s = 0; s < steps; s++) {
    const v4i a = r[2 * (size_t)idx];
    if (variant == 1) { acc += pb.z; pb = r[2 * (size_t)a.y + 1]; }                                  // extra load consumed one step later
    if (variant == 5) { acc += pb.z + pc.z; pb = r[2 * (size_t)a.y + 1]; pc = r[2 * (size_t)a.y + 2 * 77 + 1]; }  // two of them
    if (variant == 6) { int nx = a.x; for (int k = 0; k < 40; k++) nx = (nx ^ (nx >> 3)) + k - ((nx + k) ^ ((nx + k) >> 3)); acc += pb.z + (nx & 0); pb = r[2 * (size_t)a.y + 1]; }
    if (variant >= 2) o[s] = idx;
    int nx = a.x;
    if (variant == 3) { for (int k = 0; k < 40; k++) nx = (nx ^ (nx >> 3)) + k - ((nx + k) ^ ((nx + k) >> 3)) ; nx = a.x + (nx & 0); }
    idx = nx;
  }
  const long long t1 = wall_clock64();
  clk[blockIdx.x] = t1 - t0; out[(size_t)blockIdx.x * steps] = acc + idx;
}
int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 240, nrec = argc > 2 ? atoi(argv[2]) : 200000, steps = argc > 3 ? atoi(argv[3]) : 50000;
  std::vector<int> h((size_t)blocks * nrec * 8);
  std::mt19937 rng(1);
  for (int b = 0; b < blocks; b++) {
    std::vector<int> perm(nrec); std::iota(perm.begin(), perm.end(), 0); std::shuffle(perm.begin(), perm.end(), rng);
    for (int i = 0; i < nrec; i++) { int *p = &h[((size_t)b * nrec + perm[i]) * 8]; p[0] = perm[(i + 1) % nrec]; p[1] = (int)(rng() % (nrec - 100)); p[2] = 1; }
  }
  int4 *d; int *out; long long *clk;
  hipMalloc(&d, h.size() * 4); hipMalloc(&out, (size_t)blocks * steps * 4); hipMalloc(&clk, blocks * 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int variant = 0; variant <= 12; variant++) {
    const int v = variant == 4 ? 0 : variant;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 1024, 0, d, out, clk, nrec, steps, v, variant == 4); hipDeviceSynchronize(); }
    std::vector<long long> c(blocks); hipMemcpy(c.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0, mx = 0; for (auto x : c) { avg += (double)x; mx = std::max(mx, (double)x); }
    avg /= blocks;
    { int o0 = 0; hipMemcpy(&o0, out, 4, hipMemcpyDeviceToHost); printf("  [final idx of block 0: %d] ", o0); }
    printf("blocks=%d nrec=%d (%.1f MB/region) variant=%d: avg %.1f ns/step, max %.1f ns/step\n", blocks, nrec, nrec * 32.0 / 1e6, variant, avg * 10.0 / steps, mx * 10.0 / steps);
  }
  return 0;
}
