// Scheduler of the lock-step groups (lockstep.h): stackful contexts of the calling thread, recorded launches, grouped issue.
#include "lockstep.h"

#include <ucontext.h>

#include <atomic>
#include <chrono>
#include <dlfcn.h>

#include <cstdio>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace rdm {
namespace {

enum class State { Ready, Launch, Align, Sync, Done };

struct Context {
  ucontext_t uc;
  char* stack = nullptr;  // (the thread's cached stack k, below)
  State state = State::Ready;
  LaunchRecord rec;
  int rc = 0;
  std::vector<std::pair<hipEvent_t, hipStream_t>> events;  // lockstep_event: waiting for the context's next launch
  void flush_events() {
    for (auto& ev : events) (void)hipEventRecord(ev.first, ev.second);
    events.clear();
  }
};

struct Group {
  ucontext_t main;
  std::vector<Context> ctx;
  int cur = -1;
  int (*fn)(int, void*) = nullptr;
  void* user = nullptr;
};

thread_local Group* g_group = nullptr;
constexpr size_t kStackBytes = size_t(1) << 20;
struct Stacks {  // a host thread's context stacks, kept between runs (a fresh zero-filled megabyte per pair and run costs more than the switches)
  char* s[kGroupMax] = {};
  ~Stacks() {
    for (char* p : s) delete[] p;
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

void trampoline() {
  Group* g = g_group;
  Context& c = g->ctx[g->cur];
  c.rc = g->fn(g->cur, g->user);
  c.state = State::Done;
  swapcontext(&c.uc, &g->main);
}

void yield_to_scheduler(State s) {
  Group* g = g_group;
  Context& c = g->ctx[g->cur];
  c.state = s;
  swapcontext(&c.uc, &g->main);
}

}  // namespace

bool lockstep_active() { return g_group != nullptr && g_group->cur >= 0; }

void lockstep_submit(const LaunchRecord& rec) {
  g_group->ctx[g_group->cur].rec = rec;
  yield_to_scheduler(State::Launch);
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
    Context& c = g.ctx[k];
    if (!g_stacks.s[k]) g_stacks.s[k] = new char[kStackBytes];
    c.stack = g_stacks.s[k];
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = c.stack;
    c.uc.uc_stack.ss_size = kStackBytes;
    c.uc.uc_link = &g.main;
    makecontext(&c.uc, trampoline, 0);
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
      swapcontext(&g.main, &c.uc);
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
      g.ctx[k].rec.fire(recs, m);
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
