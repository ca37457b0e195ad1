// Scheduler of the lock-step groups (lockstep.h): stackful contexts of the calling thread, recorded launches, grouped issue.
#include "lockstep.h"

#include <ucontext.h>

#include <atomic>
#include <chrono>
#include <utility>
#include <vector>

namespace rdm {
namespace {

enum class State { Ready, Launch, Sync, Done };

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
namespace {
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
    // issue what was recorded: consecutive contexts with the same kernel (and dynamic LDS size) as one launch
    int launched = 0;
    for (int k = 0; k < n;) {
      if (g.ctx[k].state != State::Launch) {
        ++k;
        continue;
      }
      const LaunchRecord* recs[kGroupMax];
      int m = 0, j = k;
      for (; j < n; ++j) {
        if (g.ctx[j].state != State::Launch) continue;  // (contexts that wait or ended do not break a group)
        if (g.ctx[j].rec.fire != g.ctx[k].rec.fire || g.ctx[j].rec.lds != g.ctx[k].rec.lds) break;
        recs[m++] = &g.ctx[j].rec;
        g.ctx[j].flush_events();
      }
      g.ctx[k].rec.fire(recs, m);
      g_stats[3] += 1;
      g_stats[4] += m;
      for (int i = k; i < j; ++i)
        if (g.ctx[i].state == State::Launch) g.ctx[i].state = State::Ready;
      launched += m;
      k = j;
    }
    if (launched > 0) continue;
    // nothing to issue: the contexts still alive all wait for the stream
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
extern "C" void rdm_lockstep_stats(long long* out, int reset) {
  for (int i = 0; i < 8; ++i) {
    out[i] = rdm::g_stats[i].load();
    if (reset) rdm::g_stats[i] = 0;
  }
}
