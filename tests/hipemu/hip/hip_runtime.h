// tests/hipemu/hip/hip_runtime.h — TEST INFRASTRUCTURE ONLY.
//
// A minimal "HIP on host threads" shim so that the product's .hip sources can be compiled with g++
// and their *logic* exercised in this GPU-less container (`pytest -m "not gpu"`).  It is never part
// of the product: libuvolcodec.so is built by hipcc for gfx950 only and refuses to run without a
// GPU.  The shim runs workgroups on a few OS threads, every GPU thread being a fiber (so __syncthreads, LDS and the
// wave-level intrinsics behave), which is enough to debug kernels before spending GPU minutes.
//
// Restrictions a kernel must respect to run here (all are good gfx950 practice anyway):
//  * every lane of a wave reaches every wave intrinsic (__shfl*, __ballot, ...), no early return;
//  * every thread of a block reaches every __syncthreads();
//  * wave size is 64.
#pragma once
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__ __restrict

struct dim3 { unsigned x, y, z; dim3(unsigned X = 1, unsigned Y = 1, unsigned Z = 1) : x(X), y(Y), z(Z) {} };
struct hipemu_uint3 { unsigned x, y, z; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }
struct alignas(16) int4 { int x, y, z, w; };
inline int4 make_int4(int x, int y, int z, int w) { int4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
extern thread_local hipemu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;
extern thread_local char *hipemu_dyn_smem;
static const int warpSize = 64;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNoDevice = 100 };
typedef struct hipemu_stream *hipStream_t;
typedef struct hipemu_event { double t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// HIPEMU_DEVICES=N: the emulation reports N devices (all the same heap), so that the multi-device HOST paths - `uvolenc --gpus N`, one
// thread and one context pair per device - run without a GPU
inline hipError_t hipGetDeviceCount(int *n) { static const int nd = [] { const char *e = getenv("HIPEMU_DEVICES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 64 ? 64 : v); }(); *n = nd; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "hipemu"); strcpy(p->gcnArchName, "gfx950-hipemu"); p->multiProcessorCount = 256; p->totalGlobalMem = (size_t)8 << 30; return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
double hipemu_now_ms();
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event{0}; return hipSuccess; }
#define hipEventDisableTiming 2
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = hipemu_now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0

// ---- block execution engine ----
void hipemu_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
void hipemu_syncthreads();
// wave collectives: exchange one 64-bit value per lane
unsigned long long hipemu_wave_exchange(unsigned long long v, int src_lane, bool *valid, int width = 64);
unsigned long long hipemu_wave_ballot(int pred);

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu_launch(dim3(grid), dim3(block), (shmem), [&]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { hipemu_syncthreads(); }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }

inline int hipemu_lane() { return (int)((threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) & 63); }
template <class T> inline T __shfl(T v, int src, int width = 64) {
  unsigned long long u = 0; memcpy(&u, &v, sizeof(T));
  int lane = hipemu_lane(); int base = lane & ~(width - 1);
  bool ok; unsigned long long r = hipemu_wave_exchange(u, base + (src & (width - 1)), &ok, width);
  T o; memcpy(&o, &r, sizeof(T)); return ok ? o : v;
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  unsigned long long u = 0; memcpy(&u, &v, sizeof(T));
  int lane = hipemu_lane(); int src = lane + (int)d; bool inr = (src & ~(width - 1)) == (lane & ~(width - 1));
  bool ok; unsigned long long r = hipemu_wave_exchange(u, inr ? src : lane, &ok);
  T o; memcpy(&o, &r, sizeof(T)); return (inr && ok) ? o : v;
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  unsigned long long u = 0; memcpy(&u, &v, sizeof(T));
  int lane = hipemu_lane(); int src = lane - (int)d; bool inr = src >= 0 && (src & ~(width - 1)) == (lane & ~(width - 1));
  bool ok; unsigned long long r = hipemu_wave_exchange(u, inr ? src : lane, &ok);
  T o; memcpy(&o, &r, sizeof(T)); return (inr && ok) ? o : v;
}
template <class T> inline T __shfl_xor(T v, int m, int width = 64) {
  unsigned long long u = 0; memcpy(&u, &v, sizeof(T));
  int lane = hipemu_lane(); bool ok; unsigned long long r = hipemu_wave_exchange(u, lane ^ m, &ok, width);
  T o; memcpy(&o, &r, sizeof(T)); return ok ? o : v;
}
inline unsigned long long __ballot(int p) { return hipemu_wave_ballot(p); }
inline int __any(int p) { return hipemu_wave_ballot(p) != 0; }
inline int __all(int p) { return hipemu_wave_ballot(!p) == 0; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fdiv_rn(float a, float b) { return a / b; }

template <class T> inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicSub(T *p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicCAS(T *p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return cmp; }
template <class T> inline T atomicMin(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
template <class T> inline T atomicMax(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
