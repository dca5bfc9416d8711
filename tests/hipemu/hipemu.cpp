// tests/hipemu/hipemu.cpp — TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h in this directory).
// Executes a kernel launch on the host: workgroups are distributed over a few OS worker threads;
// inside a workgroup every GPU thread is a ucontext fiber, scheduled round-robin and switched only
// at __syncthreads() / wave intrinsics.  `__shared__` maps to `static thread_local`, so each worker
// (= one resident workgroup) owns its LDS.
#include "hip/hip_runtime.h"
#include <chrono>
#include <ucontext.h>

thread_local hipemu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
thread_local char *hipemu_dyn_smem = nullptr;

double hipemu_now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

namespace {
constexpr int MAXT = 1024;
constexpr size_t STACK = 256 * 1024;

struct Fiber { ucontext_t ctx; char *stack = nullptr; bool done = true; hipemu_uint3 tid; };

struct BlockExec {
  Fiber fib[MAXT];
  ucontext_t sched;
  int nthreads = 0, cur = 0;
  // block barrier + per-wave barriers (counter/generation: independent of fiber scheduling order)
  int arrived = 0; unsigned generation = 0;
  int warr[MAXT / 64] = {0}; unsigned wgen[MAXT / 64] = {0};
  int garr[MAXT / 16] = {0}; unsigned ggen[MAXT / 16] = {0};      // 16-lane sub-group rendezvous (shuffles with width 16 under group-divergent control flow)
  // wave exchange
  unsigned long long slots[MAXT]; int preds[MAXT];
  const std::function<void()> *body = nullptr;
  dim3 grid, block; hipemu_uint3 bidx;
  std::vector<char> dyn;
};
thread_local BlockExec *tl_exec = nullptr;

void fiber_entry() {
  BlockExec *E = tl_exec;
  Fiber &F = E->fib[E->cur];
  threadIdx = F.tid;
  (*E->body)();
  F.done = true;
  swapcontext(&F.ctx, &E->sched);
}

inline void yield_to_sched() {
  BlockExec *E = tl_exec;
  Fiber &F = E->fib[E->cur];
  swapcontext(&F.ctx, &E->sched);
  threadIdx = F.tid;          // scheduler clobbers the thread-locals when it runs other fibers
}

void run_block(BlockExec *E) {
  const int n = E->nthreads;
  blockDim = E->block; gridDim = E->grid; blockIdx = E->bidx;
  hipemu_dyn_smem = E->dyn.data() + ((64 - ((uintptr_t)E->dyn.data() & 63)) & 63);
  if (n == 1) { threadIdx = {0, 0, 0}; (*E->body)(); return; }
  E->arrived = 0;
  for (int w = 0; w < MAXT / 64; w++) E->warr[w] = 0;
  for (int w = 0; w < MAXT / 16; w++) E->garr[w] = 0;
  for (int t = 0; t < n; t++) {
    Fiber &F = E->fib[t];
    if (!F.stack) F.stack = (char *)malloc(STACK);
    getcontext(&F.ctx);
    F.ctx.uc_stack.ss_sp = F.stack; F.ctx.uc_stack.ss_size = STACK; F.ctx.uc_link = nullptr;
    makecontext(&F.ctx, fiber_entry, 0);
    F.done = false;
    F.tid.x = t % E->block.x; F.tid.y = (t / E->block.x) % E->block.y; F.tid.z = t / (E->block.x * E->block.y);
  }
  int live = n;
  while (live > 0) {
    live = 0;
    for (int t = 0; t < n; t++) {
      Fiber &F = E->fib[t];
      if (F.done) continue;
      E->cur = t;
      swapcontext(&E->sched, &F.ctx);
      if (!F.done) live++;
    }
  }
}
int flat_tid() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)); }
}  // namespace

void hipemu_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads > MAXT || nthreads <= 0) { fprintf(stderr, "hipemu: bad block size %d\n", nthreads); abort(); }
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  // one kernel at a time: several host threads (e.g. uvolenc's geometry and texture stages) share the helper pool
  static std::mutex launch_mu; std::lock_guard<std::mutex> launch_guard(launch_mu);
  static int ncores = [] { int n = (int)std::thread::hardware_concurrency(); const char *e = getenv("HIPEMU_THREADS"); if (e) n = atoi(e); return n < 1 ? 1 : (n > 16 ? 16 : n); }();
  const int nworkers = (int)std::min<size_t>((size_t)ncores, nblocks);
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    static thread_local BlockExec *E = nullptr;
    if (!E) E = new BlockExec();
    tl_exec = E;
    E->body = &body; E->grid = grid; E->block = block; E->nthreads = nthreads;
    if (E->dyn.size() < shmem + 64) E->dyn.resize(shmem + 64);
    for (;;) {
      size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      E->bidx.x = (unsigned)(b % grid.x); E->bidx.y = (unsigned)((b / grid.x) % grid.y); E->bidx.z = (unsigned)(b / ((size_t)grid.x * grid.y));
      run_block(E);
    }
  };
  if (nworkers <= 1) { worker(); return; }
  // persistent helper pool
  struct Pool {
    std::vector<std::thread> th; std::mutex m; std::condition_variable cv, cvd; const std::function<void()> *job = nullptr;
    unsigned epoch = 0; int want = 0, done = 0; bool quit = false;
    void loop(int id) { unsigned seen = 0; for (;;) { { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return quit || (epoch != seen && id < want); }); if (quit) return; seen = epoch; } (*job)(); { std::unique_lock<std::mutex> lk(m); if (++done == want) cvd.notify_all(); } } }
  };
  static Pool *P = new Pool();
  while ((int)P->th.size() < nworkers - 1) { int id = (int)P->th.size(); P->th.emplace_back([id] { P->loop(id); }); }
  std::function<void()> job = worker;
  { std::unique_lock<std::mutex> lk(P->m); P->job = &job; P->want = nworkers - 1; P->done = 0; P->epoch++; }
  P->cv.notify_all();
  worker();
  { std::unique_lock<std::mutex> lk(P->m); P->cvd.wait(lk, [&] { return P->done == P->want; }); P->want = 0; }
}

void hipemu_syncthreads() {
  BlockExec *E = tl_exec;
  if (!E || E->nthreads == 1) return;
  unsigned g = E->generation;
  if (++E->arrived == E->nthreads) { E->arrived = 0; E->generation++; return; }
  while (E->generation == g) yield_to_sched();
}

static inline void wave_barrier(BlockExec *E, int w, int wn) {
  unsigned g = E->wgen[w];
  if (++E->warr[w] == wn) { E->warr[w] = 0; E->wgen[w]++; return; }
  while (E->wgen[w] == g) yield_to_sched();
}

static inline void group_barrier(BlockExec *E, int g, int gn) {
  unsigned gen = E->ggen[g];
  if (++E->garr[g] == gn) { E->garr[g] = 0; E->ggen[g]++; return; }
  while (E->ggen[g] == gen) yield_to_sched();
}
// width 64: the whole wave meets; width 16: only the 16-lane group of the caller does (the groups of a wave may be in different
// branches, as on the GPU, where a width-16 shuffle only needs its own 16 source lanes to be active)
unsigned long long hipemu_wave_exchange(unsigned long long v, int src_lane, bool *valid, int width) {
  BlockExec *E = tl_exec;
  if (!E || E->nthreads == 1) { *valid = src_lane == 0; return v; }
  const int tid = flat_tid(), n = E->nthreads, w = tid / 64, wn = std::min(64, n - w * 64);
  E->slots[tid] = v;
  if (width == 16) {
    const int g = tid / 16, gn = std::min(16, n - g * 16);
    group_barrier(E, g, gn);
    const bool ok = src_lane >= 0 && src_lane < wn;
    const unsigned long long r = ok ? E->slots[w * 64 + src_lane] : v;
    group_barrier(E, g, gn);
    *valid = ok; return r;
  }
  wave_barrier(E, w, wn);
  const bool ok = src_lane >= 0 && src_lane < wn;
  const unsigned long long r = ok ? E->slots[w * 64 + src_lane] : v;
  wave_barrier(E, w, wn);
  *valid = ok; return r;
}

unsigned long long hipemu_wave_ballot(int pred) {
  BlockExec *E = tl_exec;
  if (!E || E->nthreads == 1) return pred ? 1ull : 0ull;
  const int tid = flat_tid(), n = E->nthreads, w = tid / 64, wn = std::min(64, n - w * 64);
  E->preds[tid] = pred != 0;
  wave_barrier(E, w, wn);
  unsigned long long m = 0;
  for (int i = 0; i < wn; i++) if (E->preds[w * 64 + i]) m |= 1ull << i;
  wave_barrier(E, w, wn);
  return m;
}
