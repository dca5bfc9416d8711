// pmccal.hip — diagnostic (not product): kernels with KNOWN HBM byte counts, run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`
// to calibrate the counters for the access patterns of this codec (MI355X_MICROARCH.md: FETCH_SIZE reports half of a 16-byte-per-lane
// streaming read on gfx950; other widths and WRITE_SIZE are uncalibrated).  Buffers are 2 GiB (>> the 256 MiB memory-side cache).
//   cal_stream16_read   every lane reads 16 consecutive bytes        requested = buffer size
//   cal_stream4_read    every lane reads 4 consecutive bytes         requested = buffer size
//   cal_gather8_read    every lane reads 8 bytes at a random 32-byte-aligned place (the walkers' record fetch): one 64-byte sector /
//                       one 128-byte line per lane is the least the memory system can move
//   cal_stream16_write, cal_scatter4_write (4 bytes at a random 32-byte-aligned place: the walkers' flag / bitmap stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void cal_stream16_read(const uint4 *p, size_t n, uint32_t *sink) { uint32_t a = 0; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; a += v.x ^ v.y ^ v.z ^ v.w; } if (a == 0x12345678u) *sink = a; }
__global__ void cal_stream4_read(const uint32_t *p, size_t n, uint32_t *sink) { uint32_t a = 0; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i]; if (a == 0x12345678u) *sink = a; }
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void cal_gather8_read(const uint2 *p, size_t n32, size_t count, uint32_t *sink) { uint32_t a = 0; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) { const uint2 v = p[4 * (mix(i) % n32)]; a += v.x ^ v.y; } if (a == 0x12345678u) *sink = a; }
__global__ void cal_stream16_write(uint4 *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3); }
__global__ void cal_scatter4_write(uint32_t *p, size_t n32, size_t count) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) p[8 * (mix(i) % n32)] = (uint32_t)i; }
int main() {
  const size_t bytes = (size_t)2 << 30, count = (size_t)64 << 20;
  void *a; uint32_t *sink; hipMalloc(&a, bytes); hipMalloc(&sink, 4); hipMemset(a, 1, bytes);
  hipLaunchKernelGGL(cal_stream16_read, dim3(4096), dim3(256), 0, 0, (const uint4 *)a, bytes / 16, sink);
  hipLaunchKernelGGL(cal_stream4_read, dim3(4096), dim3(256), 0, 0, (const uint32_t *)a, bytes / 4, sink);
  hipLaunchKernelGGL(cal_gather8_read, dim3(4096), dim3(256), 0, 0, (const uint2 *)a, bytes / 32, count, sink);
  hipLaunchKernelGGL(cal_stream16_write, dim3(4096), dim3(256), 0, 0, (uint4 *)a, bytes / 16);
  hipLaunchKernelGGL(cal_scatter4_write, dim3(4096), dim3(256), 0, 0, (uint32_t *)a, bytes / 32, count);
  hipDeviceSynchronize();
  printf("{\"buffer_bytes\": %zu, \"gather_count\": %zu, \"requested\": {\"cal_stream16_read\": %zu, \"cal_stream4_read\": %zu, \"cal_gather8_read\": %zu, \"cal_stream16_write\": %zu, \"cal_scatter4_write\": %zu}}\n",
         bytes, count, bytes, bytes, count * 8, bytes, count * 4);
  return 0;
}
