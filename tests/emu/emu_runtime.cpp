// emu_runtime.cpp -- the fiber scheduler behind tests/emu/include/hip/hip_runtime.h.  TEST INFRASTRUCTURE (see that header).
// One OS thread runs one workgroup at a time; each GPU thread is a fiber with its own stack.  A fiber runs until it reaches a
// meeting point (cross-lane operation, wavefront barrier, workgroup barrier) or returns; when every live lane of a wavefront
// waits at the same kind of point the scheduler resolves it (computes the cross-lane results with the ISA's rules) and lets the
// lanes go on.  Lanes of one wavefront waiting at different kinds of points are reported: on the GPU that is divergent control
// flow around a cross-lane operation, which the kernels do not rely on.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <thread>
#include <vector>
#include <atomic>
#include <mutex>
#include <execinfo.h>
#include <dlfcn.h>
#include <ucontext.h>
#include <signal.h>
#include <unistd.h>

thread_local emu::Idx3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
alignas(64) thread_local uint8_t smem[160 * 1024];
alignas(64) thread_local uint8_t k4_smem[160 * 1024];

void *emu_alloc(size_t n) {
  // anonymous zero pages: multi-gigabyte arenas cost nothing until touched
  if (n == 0) n = 1;
  const size_t total = n + 64;
  void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (p == MAP_FAILED) return nullptr;
  *(size_t *)p = total;
  return (uint8_t *)p + 64;
}
// ---- several emulated devices (MI_EMU_DEVICES): which device an allocation belongs to, checked wherever the host names device memory ----
#include <map>
#include <mutex>
static std::mutex g_reg_mu;
static std::map<uintptr_t, std::pair<size_t, int>> g_reg;          // base -> (bytes, device)
static std::atomic<long> g_launches[64];
namespace emu {
int device_count() { static const int n = [] { const char *e = getenv("MI_EMU_DEVICES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 64 ? 64 : v); }(); return n; }
int &cur_dev() { static thread_local int d = 0; return d; }
void note_launch() { g_launches[cur_dev()]++; }
void check_dev_ptr(const void *p, const char *what) {
  if (!p || device_count() == 1) return;
  std::lock_guard<std::mutex> lk(g_reg_mu);
  auto it = g_reg.upper_bound((uintptr_t)p);
  if (it == g_reg.begin()) return;
  --it;
  if ((uintptr_t)p >= it->first + it->second.first) return;       // host memory
  if (it->second.second != cur_dev()) {
    fprintf(stderr, "emu: %s names memory of device %d while the thread's current device is %d\n", what, it->second.second, cur_dev());
    abort();
  }
}
}
extern "C" long emu_launch_count(int dev) { return dev >= 0 && dev < 64 ? g_launches[dev].load() : -1; }
void *emu_alloc_dev(size_t n) {
  void *p = emu_alloc(n);
  if (p) { std::lock_guard<std::mutex> lk(g_reg_mu); g_reg[(uintptr_t)p] = { n ? n : 1, emu::cur_dev() }; }
  return p;
}
void emu_free(void *p) {
  if (!p) return;
  { std::lock_guard<std::mutex> lk(g_reg_mu); g_reg.erase((uintptr_t)p); }
  void *b = (uint8_t *)p - 64; munmap(b, *(size_t *)b);
}
double emu_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
)");

namespace emu {
enum Wait { W_RUN = 0, W_XLANE, W_WAVE, W_WG, W_DONE, W_SLEEP };
struct XArgs { int kind, val, old, p0, rm, bm, bc, site; };
struct Fiber { void *sp; uint8_t *stack; int wait; XArgs x; int out; };
static const size_t STACK_BYTES = 1 << 20;   // untouched pages stay uncommitted
struct Worker {
  std::vector<Fiber> fib;                      // pool, grows to the largest block
  void *sched_sp = nullptr;
  int cur = -1, nthreads = 0;
  const std::function<void()> *body = nullptr;
  unsigned long long tick = 0;
};
static thread_local Worker W;

static void fiber_main() {
  Worker &w = W;
  (*w.body)();
  Fiber &f = w.fib[w.cur];
  f.wait = W_DONE;
  emu_switch(&f.sp, w.sched_sp);
  abort();                                     // a finished fiber is never resumed
}
static void fiber_init(Fiber &f) {
  if (!f.stack) {
    f.stack = (uint8_t *)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (f.stack == (uint8_t *)MAP_FAILED) { fprintf(stderr, "emu: cannot map a fiber stack\n"); abort(); }
  }
  // initial frame: six callee-saved registers, then the return address into fiber_main; at fiber_main's entry rsp % 16 == 8
  uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
  void **sp = (void **)(top - 8);              // slot that a `call` would have pushed the return address into
  *sp = nullptr;                               // fake return address of fiber_main (never used)
  *--sp = (void *)fiber_main;                  // popped by emu_switch's ret
  for (int i = 0; i < 6; i++) *--sp = nullptr;
  f.sp = sp; f.wait = W_RUN;
}
static inline void yield_to_scheduler(int why) {
  Worker &w = W;
  Fiber &f = w.fib[w.cur];
  f.wait = why;
  emu_switch(&f.sp, w.sched_sp);
}
int xlane(int kind, int val, int old, int p0, int rm, int bm, int bc, int site) {
  Worker &w = W;
  Fiber &f = w.fib[w.cur];
  f.x = XArgs{ kind, val, old, p0, rm, bm, bc, site };
  yield_to_scheduler(W_XLANE);
  return w.fib[w.cur].out;
}
void wave_barrier() { yield_to_scheduler(W_WAVE); }
void wg_barrier() { yield_to_scheduler(W_WG); }
unsigned long long ticks() { return ++W.tick; }
// A workgroup waiting for another one (the tile search's row workers).  The one it waits for runs on another OS thread of the pool; if it has not
// moved after a very long time the pool is too small for the launch (MI_EMU_THREADS below the workers per tile) or the protocol is stuck.
static thread_local unsigned long long spins = 0;
void spin_yield() {
  // inside a workgroup the lane steps aside so that the other wavefronts of its own workgroup can run too (a wave polling an LDS flag another wave sets)
  if (W.cur >= 0 && W.nthreads > 64) yield_to_scheduler(W_SLEEP);
  if (++spins > 400000000ull) { fprintf(stderr, "emu: block %u has been waiting for another workgroup for too long (MI_EMU_THREADS too small for this launch?)\n", blockIdx.x); abort(); }
  std::this_thread::yield();
}

// source lane of a DPP control for `lane`, -1 when there is none (CDNA3 ISA 12.x "DPP")
static int dpp_src(int ctrl, int lane) {
  const int row = lane & ~15, r = lane & 15;
  if (ctrl >= 0x000 && ctrl <= 0x0FF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);          // quad_perm
  if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; return r + n < 16 ? lane + n : -1; }   // row_shl
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; return r >= n ? lane - n : -1; }       // row_shr
  if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; return row | ((r - n) & 15); }         // row_ror
  if (ctrl == 0x140) return row | (15 - r);                                                              // row_mirror
  if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));                                              // row_half_mirror
  if (ctrl == 0x142) return lane >= 16 ? row - 1 : -1;                                                   // row_bcast:15
  if (ctrl == 0x143) return lane >= 32 ? 31 : -1;                                                        // row_bcast:31
  fprintf(stderr, "emu: DPP control 0x%x is not modelled\n", ctrl); abort();
}
// One wavefront, lanes base .. base+n-1.  Resolves every cross-lane group (lanes waiting at the same source-level site) that is
// closed: each lane's source lane waits at that site too.  `force`: nothing else in the workgroup can move -- the first open
// group is resolved with its absent source lanes treated as inactive (DPP: `old` / 0 by bound_ctrl; readlane: 0).
static bool resolve_wave(Worker &w, int base, int n, bool force) {
  bool progress = false;
  int nlive = 0, nwave = 0;
  for (int l = 0; l < n; l++) { const int s = w.fib[base + l].wait; nlive += s != W_DONE; nwave += s == W_WAVE; }
  if (nlive && nwave == nlive) { for (int l = 0; l < n; l++) if (w.fib[base + l].wait == W_WAVE) w.fib[base + l].wait = W_RUN; return true; }
  bool seen[64] = { false };
  for (int l0 = 0; l0 < n; l0++) {
    if (w.fib[base + l0].wait != W_XLANE || seen[l0]) continue;
    const int site = w.fib[base + l0].x.site, kind = w.fib[base + l0].x.kind;
    bool in[64]; int cnt = 0;
    for (int l = 0; l < n; l++) { in[l] = w.fib[base + l].wait == W_XLANE && w.fib[base + l].x.site == site && w.fib[base + l].x.kind == kind; if (in[l]) { seen[l] = true; cnt++; } }
    int src[64]; bool closed = true;
    for (int l = 0; l < n; l++) {
      if (!in[l]) continue;
      const XArgs &a = w.fib[base + l].x;
      int s = -1;
      if (kind == X_READLANE || kind == X_SHFL) s = a.p0 & 63;
      else if (kind == X_DPP) { const bool en = ((a.rm >> (l >> 4)) & 1) && ((a.bm >> ((l >> 2) & 3)) & 1); s = en ? dpp_src(a.p0, l) : -2; }
      else if (kind == X_READFIRST) { s = -3; if (cnt != nlive) closed = false; }
      src[l] = s;
      if (s >= 0 && (s >= n || !in[s])) closed = false;
    }
    if (!closed && !force) continue;
    if (!closed) { static const bool trace = getenv("MI_EMU_TRACE") != nullptr; if (trace) fprintf(stderr, "emu: open cross-lane group resolved (block %u wave %d site %d kind %d, %d of %d live lanes)\n", blockIdx.x, base / 64, site, kind, cnt, nlive); }
    int first = -1; for (int l = 0; l < n; l++) if (in[l]) { first = l; break; }
    int out[64];
    for (int l = 0; l < n; l++) {
      if (!in[l]) continue;
      const XArgs &a = w.fib[base + l].x; const int s = src[l];
      if (s == -3) out[l] = w.fib[base + first].x.val;
      else if (s == -2) out[l] = a.old;
      else if (s >= 0 && s < n && in[s]) out[l] = w.fib[base + s].x.val;
      else out[l] = kind == X_DPP ? (a.bc ? 0 : a.old) : 0;
    }
    for (int l = 0; l < n; l++) if (in[l]) { w.fib[base + l].out = out[l]; w.fib[base + l].wait = W_RUN; }
    progress = true;
    if (force) return true;
  }
  return progress;
}
static const bool reverse_lanes = getenv("MI_EMU_REVERSE") != nullptr;
static void run_block(Worker &w, int nthreads) {
  if ((int)w.fib.size() < nthreads) w.fib.resize(nthreads, Fiber{ nullptr, nullptr, W_DONE, {}, 0 });
  for (int t = 0; t < nthreads; t++) fiber_init(w.fib[t]);
  w.nthreads = nthreads;
  const int nwaves = (nthreads + 63) / 64;
  int done = 0;
  while (done < nthreads) {
    bool progress = false;
    for (int wi = 0; wi < nwaves; wi++) {
      const int wv = reverse_lanes ? nwaves - 1 - wi : wi;
      const int base = wv * 64, n = nthreads - base < 64 ? nthreads - base : 64;
      for (int l = 0; l < n; l++) if (w.fib[base + l].wait == W_SLEEP) w.fib[base + l].wait = W_RUN;
      for (int li = 0; li < n; li++) {              // every runnable lane goes on to its next meeting point
        const int l = reverse_lanes ? n - 1 - li : li;   // MI_EMU_REVERSE=1: a kernel whose result depends on the order in which the lanes of a
        Fiber &f = w.fib[base + l];                      // wavefront pass between two meeting points has an unsynchronised exchange
        if (f.wait != W_RUN) continue;
        w.cur = base + l; threadIdx.x = (unsigned)(base + l);
        emu_switch(&w.sched_sp, f.sp);
        progress = true;
        if (f.wait == W_DONE) done++;
      }
      if (resolve_wave(w, base, n, false)) progress = true;
    }
    int live = 0, at_wg = 0;
    for (int t = 0; t < nthreads; t++) { live += w.fib[t].wait != W_DONE; at_wg += w.fib[t].wait == W_WG; }
    if (live > 0 && at_wg == live) { for (int t = 0; t < nthreads; t++) if (w.fib[t].wait == W_WG) w.fib[t].wait = W_RUN; progress = true; }
    if (!progress) {                                // nothing can move: an open group has to go with its absent lanes inactive
      bool forced = false;
      for (int wv = 0; wv < nwaves && !forced; wv++) { const int base = wv * 64, n = nthreads - base < 64 ? nthreads - base : 64; forced = resolve_wave(w, base, n, true); }
      if (!forced) {
        for (int wv = 0; wv < nwaves && !forced; wv++) {   // lanes of a wave split between a wave barrier and something else
          const int base = wv * 64, n = nthreads - base < 64 ? nthreads - base : 64;
          for (int l = 0; l < n; l++) if (w.fib[base + l].wait == W_WAVE) { w.fib[base + l].wait = W_RUN; forced = true; }
        }
      }
      if (!forced) { fprintf(stderr, "emu: deadlock in block %u (%d of %d threads finished)\n", blockIdx.x, done, nthreads); abort(); }
      static const bool strict = getenv("MI_EMU_STRICT") != nullptr;
      if (strict) { fprintf(stderr, "emu: MI_EMU_STRICT: an open cross-lane group had to be forced in block %u\n", blockIdx.x); abort(); }
    }
  }
}
static void on_segv(int, siginfo_t *si, void *uc) {
  static char msg[256];
  Dl_info di; memset(&di, 0, sizeof di);
  void *rip = (void *)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
  dladdr(rip, &di);
  const int n = snprintf(msg, sizeof msg, "emu: SIGSEGV at address %p (thread %u of block %u), rip %p = %s+0x%lx (%s)\n", si->si_addr, threadIdx.x, blockIdx.x, rip,
                         di.dli_fname ? di.dli_fname : "?", (unsigned long)((uintptr_t)rip - (uintptr_t)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
  (void)!write(2, msg, n);
  void **sp = (void **)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RSP];
  for (int i = 0; i < 24; i++) { Dl_info d2; memset(&d2, 0, sizeof d2); if (dladdr(sp[i], &d2) && d2.dli_fname) { const int m = snprintf(msg, sizeof msg, "  stack[%d] %p = +0x%lx (%s)\n", i, sp[i], (unsigned long)((uintptr_t)sp[i] - (uintptr_t)d2.dli_fbase), d2.dli_sname ? d2.dli_sname : "?"); (void)!write(2, msg, m); } }
  void *bt[48]; const int k = backtrace(bt, 48); backtrace_symbols_fd(bt, k, 2);
  _exit(139);
}
static void install_segv_handler() {
  static std::once_flag once;
  std::call_once(once, [] {
    if (!getenv("MI_EMU_TRACE")) return;
    static uint8_t alt[1 << 16];
    stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0; sigaltstack(&ss, nullptr);
    struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = on_segv; sa.sa_flags = SA_SIGINFO | SA_ONSTACK; sigaction(SIGSEGV, &sa, nullptr);
  });
}
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body) {
  install_segv_handler();
  static const bool trace = getenv("MI_EMU_TRACE") != nullptr;
  if (trace) fprintf(stderr, "emu: launch grid %u x %u x %u, block %u, dynamic LDS %zu\n", grid.x, grid.y, grid.z, block.x, lds_bytes);
  const unsigned nthreads = block.x * block.y * block.z;
  const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
  if (block.y != 1 || block.z != 1) { fprintf(stderr, "emu: only 1-D workgroups are modelled\n"); abort(); }
  std::atomic<unsigned long long> next{ 0 };
  auto work = [&]() {
    Worker &w = W;
    w.body = &body;
    blockDim = block; gridDim = grid; threadIdx.y = threadIdx.z = 0;
    for (;;) {
      const unsigned long long b = next.fetch_add(1);
      if (b >= nblocks) break;
      blockIdx.x = (unsigned)(b % grid.x); blockIdx.y = (unsigned)((b / grid.x) % grid.y); blockIdx.z = (unsigned)(b / ((unsigned long long)grid.x * grid.y));
      // dynamic LDS: the launch asked for lds_bytes; a canary behind it catches a kernel that writes past its allocation (on the GPU: a fault
      // or silent corruption of a neighbouring workgroup's LDS)
      uint8_t *dyn[2] = { smem, k4_smem };
      if (lds_bytes > 0 && lds_bytes + 256 <= 160 * 1024) for (int q = 0; q < 2; q++) memset(dyn[q] + lds_bytes, 0xA5, 256);
      // MI_EMU_LDS_POISON=seed: a workgroup starts with garbage in its LDS, as on the GPU (the emulator's arena otherwise holds zeros or the previous workgroup's data):
      // a kernel that reads LDS it never wrote gives other bytes
      static const char *poison = getenv("MI_EMU_LDS_POISON");
      if (poison && lds_bytes > 0) { uint32_t x = (uint32_t)atoi(poison) * 2654435761u + (uint32_t)b * 40503u + 12345u; for (int q = 0; q < 2; q++) for (size_t i = 0; i < lds_bytes; i++) { x = x * 1664525u + 1013904223u; dyn[q][i] = (uint8_t)(x >> 24); } }
      spins = 0;
      run_block(w, (int)nthreads);
      if (lds_bytes > 0 && lds_bytes + 256 <= 160 * 1024) for (int q = 0; q < 2; q++) for (int i = 0; i < 256; i++) if (dyn[q][lds_bytes + i] != 0xA5) {
        fprintf(stderr, "emu: block %u wrote past its dynamic LDS allocation of %zu bytes (offset %zu)\n", blockIdx.x, lds_bytes, lds_bytes + (size_t)i); abort(); }
    }
  };
  // workgroups are claimed in block order; a kernel whose workgroups wait for earlier ones (the tile search's row workers, up to 16 per
  // tile) needs that many of them in flight, so the pool is never smaller than 32 OS threads
  static const unsigned hw = [] { const char *e = getenv("MI_EMU_THREADS"); unsigned n = e ? (unsigned)atoi(e) : std::max(32u, std::thread::hardware_concurrency()); return n ? n : 1u; }();
  const unsigned nt = (unsigned)(nblocks < hw ? nblocks : hw);
  if (nt <= 1) { work(); return; }
  std::vector<std::thread> th;
  for (unsigned i = 0; i < nt; i++) th.emplace_back(work);
  for (auto &t : th) t.join();
}
}  // namespace emu
