// Shared host/device helpers for librdmnet_hip.so (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rdm {

// Error codes returned by every C-ABI entry point (0 = ok).
enum : int {
  RDM_OK = 0,
  RDM_ERR_ARG = -1,       // bad argument (null pointer, negative size, unsupported shape)
  RDM_ERR_WORKSPACE = -2, // caller workspace too small
  RDM_ERR_HIP = -3,       // HIP runtime error (message in rdm_last_error())
  RDM_ERR_CAPACITY = -4,  // a data-dependent capacity was exceeded on the device
};

void set_error(const char* fmt, ...);

#define RDM_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::rdm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                       __LINE__);                                                        \
      return ::rdm::RDM_ERR_HIP;                                                         \
    }                                                                                    \
  } while (0)

#define RDM_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ::rdm::set_error(__VA_ARGS__);      \
      return ::rdm::RDM_ERR_ARG;          \
    }                                     \
  } while (0)

// Set by rdm::launch (lockstep.h) when the grouped launch that carried the calling context's record failed: the error text is the
// grouped launch's; the context's next launch_status() reports it (contexts of a lock-step group share their host thread, but a
// context calls launch_status right after its launch returns, before any other context runs).
inline thread_local bool t_launch_failed = false;

inline int launch_status(const char* what) {
  if (t_launch_failed) {
    t_launch_failed = false;
    return RDM_ERR_HIP;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return RDM_ERR_HIP;
  }
  return RDM_OK;
}

// Developer knobs (A/B switches and tuning overrides behind the measurements in docs/EXPERIMENTS.md 5) exist in the LAB build only
// (`make lab` -> librdmnet_hip_lab.so, compiled with -DRDM_DEV_KNOBS; tools/*.sh select it with RDM_LIB_PATH): the
// product library never reads the environment.
#ifdef RDM_DEV_KNOBS
inline const char* dev_knob(const char* name) { return getenv(name); }
#else
inline const char* dev_knob(const char*) { return nullptr; }
#endif

// Developer knob for marginal-cost measurements (tools/exp_dup.sh, lab build): RDM_DUP="<class>[:<n>]" launches every
// kernel of that class n times (default 2) in a row -- the launches are idempotent, results do not change -- so the drop in
// pairs/s is what that class costs with several pairs in flight.  Unset (and in the product build): one launch.
inline int dup_reps(const char* cls) {
  const char* e = dev_knob("RDM_DUP");
  if (!e) return 1;
  const size_t n = strlen(cls);
  if (strncmp(e, cls, n) != 0 || (e[n] != 0 && e[n] != ':')) return 1;
  const int r = e[n] == ':' ? atoi(e + n + 1) : 2;
  return r >= 1 && r <= 16 ? r : 1;  // (0 would skip the kernel and corrupt the results)
}
#define RDM_DUP_CAT2(a, b) a##b
#define RDM_DUP_CAT(a, b) RDM_DUP_CAT2(a, b)
#define RDM_DUP_LOOP(cls) \
  static const int RDM_DUP_CAT(_dup_reps_, __LINE__) = ::rdm::dup_reps(cls); \
  for (int _dup = 0; _dup < RDM_DUP_CAT(_dup_reps_, __LINE__); ++_dup)

template <typename T>
inline T ceil_div(T a, T b) {
  return (a + b - 1) / b;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Bump allocator over a caller-provided workspace.
struct Arena {
  char* base;
  size_t cap;
  size_t off = 0;
  bool ok = true;
  Arena(void* p, size_t n) : base(static_cast<char*>(p)), cap(n) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = align_up(count * sizeof(T));
    if (base == nullptr || off + bytes > cap) {
      ok = false;
      off += bytes;
      return nullptr;
    }
    T* r = reinterpret_cast<T*>(base + off);
    off += bytes;
    return r;
  }
};

#ifdef __HIPCC__
// More than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize, which is a PER-DEVICE property of a
// kernel: `done` is a per-call-site bit mask indexed by the current device, so that a process driving several GPUs sets
// it on each of them (a single `static bool` would leave the second GPU's launch failing).  Racing first calls only
// repeat the same idempotent setting.
inline hipError_t set_max_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess) return err;
  const uint64_t bit = uint64_t(1) << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (err == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return err;
}

constexpr int kWave = 64;

// element `off` of a neighbour-index table held as int64 (i32 = 0) or int32 (i32 = 1)
__device__ __forceinline__ long long ld_index(const int64_t* p, long long off, int i32) {
  return i32 ? static_cast<long long>(reinterpret_cast<const int32_t*>(p)[off]) : static_cast<long long>(p[off]);
}
__device__ __forceinline__ void st_index(int64_t* p, long long off, long long v, int i32) {
  if (i32) reinterpret_cast<int32_t*>(p)[off] = static_cast<int32_t>(v);
  else p[off] = v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), result in every lane: four VALU adds with a DPP operand
// (xor 1, xor 2 inside the quads, then the half-row and row mirrors) instead of four ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  return v;
}
// Workgroup barrier that orders LDS traffic only: global loads issued earlier stay in flight across it
// (__syncthreads() drains vmcnt as well).
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Small device-side fills and copies.  The runtime's hipMemsetAsync / hipMemcpyAsync go through blit kernels
// with extra barrier packets and host-side bookkeeping; with several pairs in flight per GPU those cost
// noticeably more than a plain kernel launch, so the hot path uses these instead.
template <typename T>
__global__ void fill_words_kernel(T* p, int64_t n, T v) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) p[i] = v;
}
static __global__ void copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
template <typename T>
inline void fill_words(T* p, int64_t n, T v, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(fill_words_kernel<T>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, p, n, v);
}
inline void copy_words(const void* src, void* dst, int n_words, hipStream_t st) {
  if (n_words > 0)
    hipLaunchKernelGGL(copy_words_kernel, dim3((n_words + 255) / 256), dim3(256), 0, st, static_cast<const uint32_t*>(src),
                       static_cast<uint32_t*>(dst), n_words);
}

// Relaxed agent-scope load: bypasses this CU's L1, so it observes L2 atomics of other waves.
template <typename T>
__device__ __forceinline__ T ld_agent(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

}  // namespace rdm
