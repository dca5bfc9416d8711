// xcd_census.hip - which XCD (and shader engine / CU) does a workgroup land on, for a stream created with a given CU mask?
// (tools only; VERDICT r4 item 3: "determine the logical-CU -> XCD mapping (a one-kernel census of XCC_ID)")
//   hipcc --offload-arch=gfx950 -O2 tools/xcd/xcd_census.hip -o tools/xcd/xcd_census
//   xcd_census                     -> JSON: for each mask pattern, workgroups seen per XCC and distinct (xcc, se, cu) slots
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <set>
#include <string>

__global__ void k_census(uint32_t *out, int spin) {
  uint32_t xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  // stay resident for a while so that the dispatcher has to spread the grid over every CU it may use
  uint64_t t0 = wall_clock64(); while (wall_clock64() - t0 < (uint64_t)spin) { }
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

static void run(const char *name, const std::vector<uint32_t> &mask, int ncu, bool first) {
  hipStream_t s;
  if (mask.empty()) { if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return; }
  else if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s{\"mask\": \"%s\", \"error\": \"hipExtStreamCreateWithCUMask failed\"}", first ? "" : ",\n", name); (void)hipGetLastError(); return; }
  const int nb = 8 * ncu; uint32_t *d; hipMalloc(&d, nb * 8); hipMemset(d, 0xff, nb * 8);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_census, dim3(nb), dim3(256), 0, s, d, 200000);     // 2 ms at 100 MHz
  { const hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", name, hipGetErrorString(e)); return; } }
  std::vector<uint32_t> h(2 * nb); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
  std::map<int, int> per_xcc; std::set<uint32_t> slots;
  for (int i = 0; i < nb; i++) { const int x = h[2 * i] & 15; per_xcc[x]++; const uint32_t id = h[2 * i + 1]; const uint32_t cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7; slots.insert((x << 16) | (se << 8) | (sh << 4) | cu); }
  printf("%s{\"mask\": \"%s\", \"workgroups\": %d, \"distinct_cus\": %zu, \"per_xcc\": {", first ? "" : ",\n", name, nb, slots.size());
  bool f = true; for (auto &kv : per_xcc) { printf("%s\"%d\": %d", f ? "" : ", ", kv.first, kv.second); f = false; }
  printf("}}");
  hipFree(d); hipStreamDestroy(s);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); const int ncu = p.multiProcessorCount; const size_t nw = (ncu + 31) / 32;
  printf("{\"cus\": %d, \"runs\": [\n", ncu);
  run("none", {}, ncu, true);
  auto mk = [&](auto pred) { std::vector<uint32_t> m(nw, 0u); for (int i = 0; i < ncu; i++) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
  for (int k = 0; k < 8; k++) { char nm[64]; snprintf(nm, sizeof nm, "i%%8==%d", k); run(nm, mk([&](int i) { return i % 8 == k; }), ncu, false); }
  for (int k = 0; k < 8; k++) { char nm[64]; snprintf(nm, sizeof nm, "i/32==%d", k); run(nm, mk([&](int i) { return i / 32 == k; }), ncu, false); }
  run("i%8<4", mk([&](int i) { return i % 8 < 4; }), ncu, false);
  run("i%8<2", mk([&](int i) { return i % 8 < 2; }), ncu, false);
  run("i%4==3", mk([&](int i) { return i % 4 == 3; }), ncu, false);
  run("i<128", mk([&](int i) { return i < 128; }), ncu, false);
  printf("\n]}\n");
  return 0;
}
