// Lock-step execution of several pairs on ONE stream with grouped kernel launches (round 5).
//
// Four pairs in flight on four streams is what the chip's four compute pipes overlap; what remains of a pair at that point is
// mostly kernels that do not fill the GPU (docs/EXPERIMENTS.md 5e / 5f, profiles/r05_batch_lab.md: more rows per LAUNCH is the
// lever that is left).  Here B engines run their pairs on one stream in lock step: every engine's run is a stackful context
// (ucontext) of the calling host thread; a kernel launch that goes through rdm::launch<...>() is not issued but RECORDED, and the
// context yields; when every context has yielded, the scheduler issues the recorded launches of the same kernel as ONE launch
// -- the grouped kernel below runs the original kernel body once per recorded launch, each with its own arguments and its own
// grid, workgroup after workgroup in one 1-D grid -- and resumes the contexts.  A pair's arithmetic is untouched (same body,
// same arguments, same grid coordinates): the results are the bits of the pair's own run.  Launches that do not go through
// rdm::launch (kernels not converted yet) are issued at once, in the context's order -- a context only resumes after its recorded
// launch has been issued, so a pair's launches keep their order on the stream.  A host wait (size read-back) parks the context until
// every running context waits; the scheduler then waits for the stream once.
//
// Converting a kernel: its body becomes `__device__ void name_body(const dim3 blockIdx, const dim3 gridDim, params...)` (the two
// names shadow the built-ins, the body's text does not change), `__global__ name(params...)` calls it, and the launch site says
// rdm::launch<name_body, name, THREADS[, MIN_WAVES]>(grid, lds_bytes, stream, args...).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "common.h"

namespace rdm {

constexpr int kGroupMax = 8;  // recorded launches one grouped launch can carry (= pairs per lock-step group)

// ---- a trivially copyable argument pack (kernel parameters by value)
template <class... A> struct ArgPack;
template <> struct ArgPack<> {};
template <class H, class... T> struct ArgPack<H, T...> {
  H head;
  ArgPack<T...> tail;
};
template <class H, class... T> inline ArgPack<H, T...> make_pack(H h, T... t) {
  ArgPack<H, T...> p;
  p.head = h;
  if constexpr (sizeof...(T) > 0) p.tail = make_pack<T...>(t...);
  return p;
}
inline ArgPack<> make_pack() { return ArgPack<>{}; }

#ifdef __HIPCC__
template <class F, class... U> __device__ __forceinline__ void apply_pack(F&& f, const ArgPack<>&, U... u) { f(u...); }
template <class F, class H, class... T, class... U>
__device__ __forceinline__ void apply_pack(F&& f, const ArgPack<H, T...>& p, U... u) {
  apply_pack(static_cast<F&&>(f), p.tail, u..., p.head);
}

template <class... A> struct GroupArgs {
  int n;
  int first[kGroupMax + 1];            // prefix of workgroup counts
  unsigned gx[kGroupMax], gy[kGroupMax], gz[kGroupMax];
  ArgPack<A...> a[kGroupMax];
};

// One 1-D grid carrying the grids of g.n recorded launches: workgroup -> (launch, its block coordinates).
template <auto Body, int THREADS, int MINW, class... A>
__global__ __launch_bounds__(THREADS, MINW) void grouped_kernel(GroupArgs<A...> g) {
  int p = 0;
#pragma unroll
  for (int k = 1; k < kGroupMax; ++k) p += (k < g.n && static_cast<int>(blockIdx.x) >= g.first[k]) ? 1 : 0;
  const unsigned l = blockIdx.x - static_cast<unsigned>(g.first[p]);
  const unsigned gx = g.gx[p], gy = g.gy[p];
  const dim3 bid(l % gx, (l / gx) % gy, l / (gx * gy));
  const dim3 gd(gx, gy, g.gz[p]);
  apply_pack([&](auto... x) { Body(bid, gd, x...); }, g.a[p]);
}
#endif

// ---- host side
struct LaunchRecord {
  // issues `n` recorded launches (same `fire`) as one, with the largest of their dynamic LDS sizes: n == 1 -> the original kernel
  int (*fire)(const LaunchRecord* const* recs, int n);
  dim3 grid;
  size_t lds;
  hipStream_t stream;
  alignas(16) unsigned char args[768];
};

// The calling thread's lock-step group, if one is running (lockstep.cpp)
bool lockstep_active();
int lockstep_submit(const LaunchRecord& rec);   // records the launch of the running context and yields until it has been issued; returns what issuing it returned (0 = ok)
void lockstep_sync();                           // parks the running context until the group's stream is idle
// A layer boundary of the running context (no-op outside a group): parked until every context of the group that can still run has
// reached a boundary or a wait.  Pairs of different sizes do not launch the same NUMBER of kernels per layer (a split-K product has
// a reduce kernel, a one-pass one has not): without boundaries a pair that is one launch ahead stays ahead -- and ungrouped --
// until the next host wait.  Contexts that reach different boundaries are released together all the same (no deadlock; they
// only group worse).
void lockstep_align();
// An event of the running context: recorded on `stream` right before the context's next recorded launch is issued (or before
// the group waits / the context ends) -- i.e. after everything the context has launched so far, without making it yield.
void lockstep_event(hipEvent_t ev, hipStream_t stream);
// Runs fn(0) .. fn(n - 1) as the contexts of one lock-step group on `stream`; wait(stream) is the group's host wait.
int lockstep_run(int n, int (*fn)(int, void*), void* user, hipStream_t stream, int (*wait)(hipStream_t, void*), void* wait_user,
                 int* rcs);

#ifdef __HIPCC__
template <class... P> struct Sig {};
template <class... P> Sig<P...> sig_of(void (*)(const dim3, const dim3, P...));
template <class T> struct Ident { using type = T; };
template <class F, class... U> inline void apply_host(F&& f, const ArgPack<>&, U... u) { f(u...); }
template <class F, class H, class... T, class... U> inline void apply_host(F&& f, const ArgPack<H, T...>& p, U... u) {
  apply_host(static_cast<F&&>(f), p.tail, u..., p.head);
}

template <auto Body, auto Kernel, int THREADS, int MINW, class... P>
int fire_records(const LaunchRecord* const* recs, int n) {
  if (n == 1) {
    ArgPack<P...> a;
    std::memcpy(&a, recs[0]->args, sizeof(a));
    apply_host([&](auto... x) { hipLaunchKernelGGL(Kernel, recs[0]->grid, dim3(THREADS), recs[0]->lds, recs[0]->stream, x...); }, a);
    const hipError_t lerr = hipGetLastError();
    if (lerr != hipSuccess) {
      set_error("launch (one record of a lock-step group) failed: %s", hipGetErrorString(lerr));
      return RDM_ERR_HIP;
    }
    return 0;
  }
  GroupArgs<P...> g;
  g.n = n;
  g.first[0] = 0;
  for (int k = 0; k < n; ++k) {
    const dim3 gr = recs[k]->grid;
    g.gx[k] = gr.x; g.gy[k] = gr.y; g.gz[k] = gr.z;
    g.first[k + 1] = g.first[k] + static_cast<int>(gr.x * gr.y * gr.z);
    std::memcpy(&g.a[k], recs[k]->args, sizeof(ArgPack<P...>));
  }
  for (int k = n; k < kGroupMax; ++k) {
    g.first[k + 1] = g.first[n];
    g.gx[k] = g.gy[k] = g.gz[k] = 1;
  }
  size_t lds = 0;  // (dynamic LDS is an upper bound of what a body uses: the group gets the largest request)
  for (int k = 0; k < n; ++k) lds = recs[k]->lds > lds ? recs[k]->lds : lds;
  if (lds > 32768) {  // (dynamic + static LDS beyond 64 KB needs the attribute on THIS instantiation, per device: set from 32 KB of dynamic LDS up)
    static std::atomic<uint64_t> done{0};
    const hipError_t err = set_max_dynamic_lds(reinterpret_cast<const void*>(grouped_kernel<Body, THREADS, MINW, P...>), 160 * 1024 - 4096, done);
    if (err != hipSuccess) {
      set_error("grouped launch: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(err));
      return RDM_ERR_HIP;
    }
  }
  hipLaunchKernelGGL((grouped_kernel<Body, THREADS, MINW, P...>), dim3(static_cast<unsigned>(g.first[n])), dim3(THREADS), lds,
                     recs[0]->stream, g);
  const hipError_t lerr = hipGetLastError();  // (taken here: the contexts of the group would otherwise pick it up in resume order)
  if (lerr != hipSuccess) {
    set_error("grouped launch of %d records failed: %s", n, hipGetErrorString(lerr));
    return RDM_ERR_HIP;
  }
  return 0;
}
template <auto Body, auto Kernel, int THREADS, int MINW, class... P>
inline void launch_sig(Sig<P...>, dim3 grid, size_t lds, hipStream_t st, typename Ident<P>::type... a) {
  static_assert(sizeof(ArgPack<P...>) <= sizeof(LaunchRecord::args), "kernel arguments too large for a launch record");
  static_assert(sizeof(GroupArgs<P...>) <= 4000, "a grouped launch of these arguments exceeds the kernel-argument segment");
  if (grid.x == 0 || grid.y == 0 || grid.z == 0) return;
  if (!lockstep_active()) {
    hipLaunchKernelGGL(Kernel, grid, dim3(THREADS), lds, st, a...);
    return;
  }
  LaunchRecord rec;
  rec.fire = &fire_records<Body, Kernel, THREADS, MINW, P...>;
  rec.grid = grid; rec.lds = lds; rec.stream = st;
  const ArgPack<P...> pack = make_pack<P...>(a...);
  std::memcpy(rec.args, &pack, sizeof(pack));
  if (lockstep_submit(rec) != 0) t_launch_failed = true;  // (launch_status of this context's caller reports it)
}
// The launch of a converted kernel: issued at once outside a lock-step group, recorded and grouped inside one.
template <auto Body, auto Kernel, int THREADS, int MINW = 1, class... U>
inline void launch(dim3 grid, size_t lds, hipStream_t st, U... a) {
  launch_sig<Body, Kernel, THREADS, MINW>(decltype(sig_of(Body)){}, grid, lds, st, a...);
}
#endif

}  // namespace rdm
