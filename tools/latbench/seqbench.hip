// seqbench.hip — diagnostic (not product): what bounds a one-lane walker whose dependent loads move through memory mostly
// sequentially (the edgebreaker walk over a lattice- or Morton-ordered record table: 32-byte face blocks, 4 per 128-byte line)?
// Each workgroup chases a chain idx -> rec[idx].next through its own region.  Chains: 0 = sequential (i -> i+1),
// 1 = strip-like (alternates between two rows `row` blocks apart), 2 = random cycle.
// Variants: 0 plain dependent chain; 1 + a second wave of the workgroup that reads the walker's position from LDS and touches the
// lines ahead of / around it (window of `win` lines), so that the walker's loads hit in L1 / L2; 2 = variant 0 + one global store per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define G(T) __attribute__((address_space(1))) T *
__global__ void __launch_bounds__(128) chase(const uint2 *recs, int *out, long long *clk, int nrec, int steps, int variant, int win) {
  __shared__ volatile int cur; __shared__ volatile int done;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) { cur = 0; done = 0; }
  __syncthreads();
  G(const u2) r = (G(const u2))(recs + 4 * (size_t)blockIdx.x * nrec);     // 4 uint2 (32 bytes) per record block
  if (wave == 1) {
    if (variant != 1) return;
    int base = -1 << 30; unsigned acc = 0;
    while (!done) {
      const int c = cur >> 2;                         // 128-byte line of the walker's position (4 blocks per line)
      if (c < base + win / 4 || c >= base + (3 * win) / 4) {                // walker left the middle of the touched window: move it
        base = c - win / 4;
        for (int l = lane; l < win; l += 64) { const int line = base + l; if (line >= 0 && line < nrec / 4) acc += r[16 * (size_t)line].x; }
      }
      __builtin_amdgcn_s_sleep(4);
    }
    if (acc == 0x12345678u) out[0] = (int)acc;
    return;
  }
  if (lane) return;
  int idx = 0; unsigned acc = 0;
  G(int) o = (G(int))(out + (size_t)blockIdx.x * steps);
  const long long t0 = wall_clock64();
  for (int s = 0; s < steps; s++) {
    const u2 a = r[4 * (size_t)idx];
    if (variant == 1) cur = idx;
    if (variant == 2) o[s] = idx;
    acc += a.y; idx = (int)a.x;
  }
  const long long t1 = wall_clock64();
  done = 1;
  clk[blockIdx.x] = t1 - t0; out[(size_t)blockIdx.x * steps] = (int)acc + idx;
}
int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 150, nrec = argc > 2 ? atoi(argv[2]) : 200000, steps = argc > 3 ? atoi(argv[3]) : 100000, row = 800;
  uint2 *d; int *out; long long *clk;
  hipMalloc(&d, (size_t)blocks * nrec * 32); hipMalloc(&out, (size_t)blocks * steps * 4 + 64); hipMalloc(&clk, blocks * 8);
  std::mt19937 rng(1);
  for (int chain = 0; chain < 3; chain++) {
    std::vector<unsigned> h((size_t)nrec * 8, 0);
    std::vector<int> order(nrec);
    if (chain == 0) std::iota(order.begin(), order.end(), 0);
    else if (chain == 1) { int n = 0; for (int r0 = 0; r0 + 2 * row <= nrec; r0 += 2 * row) for (int i = 0; i < row; i++) { order[n++] = r0 + i; order[n++] = r0 + row + i; } while (n < nrec) { order[n] = n; n++; } }
    else { std::iota(order.begin(), order.end(), 0); std::shuffle(order.begin(), order.end(), rng); }
    for (int i = 0; i < nrec; i++) { h[(size_t)order[i] * 8] = (unsigned)order[(i + 1) % nrec]; h[(size_t)order[i] * 8 + 1] = 1; }
    { int s0 = order[0]; std::swap(h[0], h[(size_t)s0 * 8]); }   // keep it simple: the chain starts at index 0 (order[0] == 0 for chains 0/1)
    if (chain == 2) { for (int i = 0; i < nrec; i++) { h[(size_t)order[i] * 8] = (unsigned)order[(i + 1) % nrec]; } }
    for (int b = 0; b < blocks; b++) hipMemcpy((char *)d + (size_t)b * nrec * 32, h.data(), (size_t)nrec * 32, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 3; variant++)
      for (int win : {32, 64, 128, 256}) {
        if (variant != 1 && win != 32) continue;
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(chase, dim3(blocks), dim3(128), 0, 0, d, out, clk, nrec, steps, variant, win); hipDeviceSynchronize(); }
        std::vector<long long> c(blocks); hipMemcpy(c.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0, mx = 0; for (auto x : c) { avg += (double)x; mx = std::max(mx, (double)x); }
        avg /= blocks;
        printf("chain=%d (%s) blocks=%d variant=%d win=%d lines: avg %.1f ns/step, max %.1f ns/step\n", chain, chain == 0 ? "sequential" : (chain == 1 ? "two-row strip" : "random"), blocks, variant, win, avg * 10.0 / steps, mx * 10.0 / steps);
      }
  }
  return 0;
}
