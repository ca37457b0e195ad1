// Scheduler of the lock-step groups (lockstep.h): stackful contexts of the calling thread, recorded launches, grouped issue.
#include "lockstep.h"

#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace rdm {
namespace {

enum class State { Ready, Launch, Align, Sync, Done };

// (self-test, end of this file: the scheduler's actions on scripted stand-in launches are logged instead of issued)
thread_local int* t_log = nullptr;
thread_local int t_log_n = 0, t_log_cap = 0, t_waits = 0;
void log_action(int code, int value) {
  if (t_log_n + 2 <= t_log_cap) {
    t_log[t_log_n] = code;
    t_log[t_log_n + 1] = value;
  }
  t_log_n += 2;
}
const hipStream_t kSelfTestStream = reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(1));

// ---- stackful contexts.  On x86-64 the switch is a dozen instructions of user-mode code (callee-saved registers, stack pointer,
// MXCSR / x87 control word): glibc's swapcontext also saves and restores the signal mask -- two rt_sigprocmask system calls per
// switch, ~1 000 switches per pair.  Other hosts use ucontext.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define RDM_CTX_ASM 1
extern "C" void rdm_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl rdm_ctx_switch
.type rdm_ctx_switch,@function
rdm_ctx_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  subq $8, %rsp
  stmxcsr (%rsp)
  fnstcw 4(%rsp)
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  ldmxcsr (%rsp)
  fldcw 4(%rsp)
  addq $8, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size rdm_ctx_switch,.-rdm_ctx_switch
)");
#else
#define RDM_CTX_ASM 0
#endif

struct Context {
#if RDM_CTX_ASM
  void* sp = nullptr;
#else
  ucontext_t uc;
#endif
  char* stack = nullptr;  // (the thread's cached stack k, below)
  State state = State::Ready;
  LaunchRecord rec;
  int rc = 0;
  int fire_rc = 0;        // what issuing this context's last recorded launch returned (launch_status of ITS launch)
  std::vector<std::pair<hipEvent_t, hipStream_t>> events;  // lockstep_event: waiting for the context's next launch
  void flush_events() {
    for (auto& ev : events) {
      if (ev.second == kSelfTestStream) log_action(4, static_cast<int>(reinterpret_cast<uintptr_t>(ev.first)));
      else (void)hipEventRecord(ev.first, ev.second);
    }
    events.clear();
  }
};

struct Group {
#if RDM_CTX_ASM
  void* main_sp = nullptr;
#else
  ucontext_t main;
#endif
  std::vector<Context> ctx;
  int cur = -1;
  int (*fn)(int, void*) = nullptr;
  void* user = nullptr;
};

thread_local Group* g_group = nullptr;
// A pair's whole run executes on its context's stack -- engine_run_once and every HIP runtime call below it (first-launch code
// object loading included): 8 MiB of address space each, committed lazily by the kernel, with an inaccessible guard page below
// (an overflow faults instead of corrupting the heap; ADVICE r5: round 5 used a 1 MiB `new char[]`).
constexpr size_t kStackBytes = size_t(8) << 20;
struct Stacks {  // a host thread's context stacks, kept between runs (mapping and faulting them in per pair costs more than the switches)
  char* s[kGroupMax] = {};
  size_t guard = 0;
  char* get(int k) {
    if (!s[k]) {
      guard = static_cast<size_t>(sysconf(_SC_PAGESIZE));
      void* p = mmap(nullptr, kStackBytes + guard, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_STACK, -1, 0);
      if (p == MAP_FAILED) return nullptr;
      (void)mprotect(p, guard, PROT_NONE);
      s[k] = static_cast<char*>(p) + guard;
    }
    return s[k];
  }
  ~Stacks() {
    for (char* p : s)
      if (p) munmap(p - guard, kStackBytes + guard);
  }
};
thread_local Stacks g_stacks;
}  // namespace
std::atomic<long long> g_stats[8];  // ns in run, ns in waits, waits, grouped launches, records, runs
std::mutex g_by_kernel_mu;
std::map<std::pair<const void*, size_t>, std::pair<long long, long long>> g_by_kernel;  // (fire, lds) -> launches, records (RDM_LOCKSTEP_STATS)
namespace {
const bool g_by_kernel_on = dev_knob("RDM_LOCKSTEP_STATS") != nullptr;  // (lab build only: the product library never reads the environment)
inline long long now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void switch_to_scheduler(Group* g, Context& c) {
#if RDM_CTX_ASM
  rdm_ctx_switch(&c.sp, g->main_sp);
#else
  swapcontext(&c.uc, &g->main);
#endif
}
void switch_to_context(Group* g, Context& c) {
#if RDM_CTX_ASM
  rdm_ctx_switch(&g->main_sp, c.sp);
#else
  swapcontext(&g->main, &c.uc);
#endif
}

void trampoline() {
  Group* g = g_group;
  Context& c = g->ctx[g->cur];
  try {
    c.rc = g->fn(g->cur, g->user);
  } catch (...) {  // (an exception must not unwind out of the context's first frame: the run counts as failed)
    c.rc = -2;
  }
  c.state = State::Done;
  switch_to_scheduler(g, c);  // (never resumed)
  abort();
}

void yield_to_scheduler(State s) {
  Group* g = g_group;
  Context& c = g->ctx[g->cur];
  c.state = s;
  switch_to_scheduler(g, c);
}

// a fresh context on `stack`: its first switch-in enters trampoline()
bool make_context(Group& g, Context& c, char* stack) {
  if (!stack) return false;
  c.stack = stack;
#if RDM_CTX_ASM
  (void)g;
  // the frame rdm_ctx_switch pops: {mxcsr, x87 cw}, r15 r14 r13 r12 rbx rbp, return address = trampoline, then the (never used)
  // return address of trampoline itself -- which leaves the stack pointer at 8 mod 16 on entry, as after a call
  uintptr_t top = (reinterpret_cast<uintptr_t>(stack) + kStackBytes) & ~uintptr_t(15);
  uint64_t* f = reinterpret_cast<uint64_t*>(top) - 9;
  uint32_t mxcsr = 0;
  uint16_t fpcw = 0;
  asm volatile("stmxcsr %0" : "=m"(mxcsr));
  asm volatile("fnstcw %0" : "=m"(fpcw));
  f[0] = static_cast<uint64_t>(mxcsr) | (static_cast<uint64_t>(fpcw) << 32);
  for (int i = 1; i <= 6; ++i) f[i] = 0;
  f[7] = reinterpret_cast<uint64_t>(&trampoline);
  f[8] = 0;
  c.sp = f;
#else
  getcontext(&c.uc);
  c.uc.uc_stack.ss_sp = c.stack;
  c.uc.uc_stack.ss_size = kStackBytes;
  c.uc.uc_link = &g.main;
  makecontext(&c.uc, trampoline, 0);
#endif
  return true;
}

}  // namespace

bool lockstep_active() { return g_group != nullptr && g_group->cur >= 0; }

int lockstep_submit(const LaunchRecord& rec) {
  Context& c = g_group->ctx[g_group->cur];
  c.rec = rec;
  c.fire_rc = 0;
  yield_to_scheduler(State::Launch);
  return c.fire_rc;  // (the grouped launch that carried this record)
}

void lockstep_sync() { yield_to_scheduler(State::Sync); }

void lockstep_align() {
  if (lockstep_active()) yield_to_scheduler(State::Align);
}

void lockstep_event(hipEvent_t ev, hipStream_t stream) { g_group->ctx[g_group->cur].events.emplace_back(ev, stream); }

int lockstep_run(int n, int (*fn)(int, void*), void* user, hipStream_t stream, int (*wait)(hipStream_t, void*), void* wait_user,
                 int* rcs) {
  if (n < 1 || n > kGroupMax || g_group != nullptr) return -1;
  Group g;
  g.ctx.resize(n);
  g.fn = fn;
  g.user = user;
  g_group = &g;
  for (int k = 0; k < n; ++k) {
    if (!make_context(g, g.ctx[k], g_stacks.get(k))) {
      g_group = nullptr;
      return -3;  // (no address space for a context stack)
    }
  }
  int wait_rc = 0;
  const long long t_run = now_ns();
  for (;;) {
    // one pass: every context that can run, runs until it records a launch, waits or ends
    bool any = false;
    for (int k = 0; k < n; ++k) {
      Context& c = g.ctx[k];
      if (c.state != State::Ready) continue;
      any = true;
      g.cur = k;
      switch_to_context(&g, c);
      g.cur = -1;
    }
    // issue what was recorded: the contexts that recorded the same kernel as one launch -- wherever they
    // stand in the group (every context has ONE record pending and only a context's own launches are ordered)
    int launched = 0;
    for (int k = 0; k < n; ++k) {
      if (g.ctx[k].state != State::Launch) continue;
      const LaunchRecord* recs[kGroupMax];
      int m = 0;
      for (int j = k; j < n; ++j) {
        if (g.ctx[j].state != State::Launch) continue;
        if (g.ctx[j].rec.fire != g.ctx[k].rec.fire) continue;
        recs[m++] = &g.ctx[j].rec;
        g.ctx[j].flush_events();
        g.ctx[j].state = State::Ready;
      }
      const int frc = g.ctx[k].rec.fire(recs, m);
      if (frc != 0)  // every context the launch carried learns that ITS launch failed
        for (int j = k; j < n; ++j)
          for (int q = 0; q < m; ++q)
            if (recs[q] == &g.ctx[j].rec) g.ctx[j].fire_rc = frc;
      if (g_by_kernel_on) {
        std::lock_guard<std::mutex> lk(g_by_kernel_mu);
        auto& e = g_by_kernel[{reinterpret_cast<const void*>(g.ctx[k].rec.fire), g.ctx[k].rec.lds}];
        e.first += 1;
        e.second += m;
      }
      g_stats[3] += 1;
      g_stats[4] += m;
      launched += m;
    }
    if (launched > 0) continue;
    // nothing to issue: contexts at a layer boundary go on together (before any wait: the others then reach the wait too)
    bool aligned = false;
    for (int k = 0; k < n; ++k) {
      if (g.ctx[k].state != State::Align) continue;
      g.ctx[k].state = State::Ready;
      aligned = true;
    }
    if (aligned) continue;
    // the contexts still alive all wait for the stream
    bool waiting = false;
    for (int k = 0; k < n; ++k) {
      waiting |= g.ctx[k].state == State::Sync;
      g.ctx[k].flush_events();  // (contexts that wait or have ended)
    }
    if (waiting) {
      const long long t_w = now_ns();
      const int rc = wait(stream, wait_user);
      g_stats[1] += now_ns() - t_w;
      g_stats[2] += 1;
      if (rc != 0 && wait_rc == 0) wait_rc = rc;
      for (int k = 0; k < n; ++k)
        if (g.ctx[k].state == State::Sync) g.ctx[k].state = State::Ready;
      continue;
    }
    if (!any) break;  // every context has ended
  }
  g_group = nullptr;
  g_stats[0] += now_ns() - t_run;
  g_stats[5] += 1;
  for (int k = 0; k < n; ++k) rcs[k] = g.ctx[k].rc;
  return wait_rc;
}

}  // namespace rdm

// Self-test of the scheduler without a GPU (tests/test_lockstep.py): n contexts run scripted sequences of recorded launches of two
// stand-in kernels, layer boundaries, deferred events and waits; every scheduler action is logged as (code, value):
//   1 = kernel A issued for `value` contexts, 2 = kernel B issued for `value` contexts, 3 = the group waited (value = waits so far),
//   4 = an event of context `value` was recorded.  script[k * len + i]: 1 / 2 = launch A / B, 3 = wait, 4 = layer boundary,
//   5 = event, 0 = nothing.  Returns the number of log entries (or -1).
namespace rdm {
namespace {
int fire_a(const rdm::LaunchRecord* const*, int n) { log_action(1, n); return 0; }
int fire_b(const rdm::LaunchRecord* const*, int n) { log_action(2, n); return 0; }
struct SelfTest { const int* script; int len; };
int selftest_ctx(int k, void* user) {
  const SelfTest& t = *static_cast<const SelfTest*>(user);
  for (int i = 0; i < t.len; ++i) {
    const int op = t.script[k * t.len + i];
    if (op == 1 || op == 2) {
      rdm::LaunchRecord rec{};
      rec.fire = op == 1 ? fire_a : fire_b;
      rec.grid = dim3(1);
      rdm::lockstep_submit(rec);
    } else if (op == 3) {
      rdm::lockstep_sync();
    } else if (op == 4) {
      rdm::lockstep_align();
    } else if (op == 5) {  // (a stand-in event: logged as 4, k when the scheduler records it)
      rdm::lockstep_event(reinterpret_cast<hipEvent_t>(static_cast<uintptr_t>(k)), kSelfTestStream);
    }
  }
  return 100 + k;
}
int selftest_wait(hipStream_t, void*) { log_action(3, ++t_waits); return 0; }
}  // namespace
}  // namespace rdm
extern "C" int rdm_lockstep_selftest(int n_ctx, const int* script, int len, int* log, int cap, int* rcs) {
  using namespace rdm;
  t_log = log; t_log_n = 0; t_log_cap = cap; t_waits = 0;
  SelfTest t{script, len};
  const int rc = lockstep_run(n_ctx, selftest_ctx, &t, nullptr, selftest_wait, nullptr, rcs);
  t_log = nullptr;
  return rc != 0 ? -1 : t_log_n / 2;
}

// Developer counters (tools/lockstep_lab.py): ns in lock-step runs, ns of them in host waits, waits, grouped launches, records, runs
extern "C" void rdm_lockstep_stats_dump() {  // (RDM_LOCKSTEP_STATS=1: launches and records per kernel and dynamic LDS size)
  std::lock_guard<std::mutex> lk(rdm::g_by_kernel_mu);
  for (auto& kv : rdm::g_by_kernel) {
    Dl_info info{};
    const char* name = dladdr(kv.first.first, &info) && info.dli_sname ? info.dli_sname : "?";
    std::printf("%8lld launches %8lld records  lds %6zu  %.200s\n", kv.second.first, kv.second.second, kv.first.second, name);
  }
  rdm::g_by_kernel.clear();
}
extern "C" void rdm_lockstep_stats(long long* out, int reset) {
  for (int i = 0; i < 8; ++i) {
    out[i] = rdm::g_stats[i].load();
    if (reset) rdm::g_stats[i] = 0;
  }
}
