// a1 -- voxel-grid barycentre subsampling, bit-exact with the reference INCLUDING its output order.
//
// Reference: geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-48.
// The reference emits voxels in the iteration order of a libstdc++ std::unordered_map<size_t, ...>
// filled in point order.  That order is a deterministic function of (a) the distinct voxel keys in
// first-occurrence order and (b) the container's growth schedule:
//   * inserting into an empty bucket puts the node at the FRONT of the global list, inserting into
//     an occupied bucket puts it at the front of that bucket's run  (hashtable.h
//     _M_insert_bucket_begin), and a rehash re-inserts every node in current list order with the
//     same rule (_M_rehash_aux);
//   * hence after each growth stage the list is  sort by (first-time-of-bucket desc, time desc)
//     where `time` = position in [previous list ++ newly inserted keys].
// Each stage is therefore a data-parallel regrouping (atomicMin / atomicAdd per bucket, one
// suffix scan, tiny per-bucket sorts); no serial pointer chasing is needed.  One 1024-thread
// workgroup per cloud runs all stages; the growth schedule itself comes from libstdc++'s own
// _Prime_rehash_policy on the host (rdm_rehash_schedule), so it follows the installed library.
//
// Float semantics (must not be contracted): origin = floor(min * (float)(1/v)) * v,
// i = floor((p - origin) / v) with an IEEE fp32 divide, sums are sequential fp32 adds in point
// order, output = sum * (float)(1.0 / count).
#pragma clang fp contract(off)

#include <atomic>
#include <cstdlib>
#include <unordered_map>  // std::__detail::_Prime_rehash_policy

#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "internal.h"

namespace {

using namespace rdm;

constexpr int kT = 1024;
// LDS-resident replay of the growth stages (P7) for clouds of up to kLdsM voxels / kLdsB buckets: the order lists
// and the per-bucket tables then live in 144 KB of the CU's 160 KB LDS instead of HBM (one workgroup per CU)
constexpr int kLdsM = 10304, kLdsB = 10304;
constexpr int kOwn = (kLdsM + kT - 1) / kT;  // stage elements owned by one thread
constexpr size_t kLdsBytes = 3 * sizeof(unsigned short) * kLdsM + 2 * sizeof(unsigned) * kLdsB;
constexpr int kMaxStages = 40;
constexpr int64_t kMultiMinPoints = 16384;  // stacked points from which rdm_grid_subsample uses the multi-launch form
constexpr unsigned long long kEmpty = ~0ull;

struct Schedule {
  int n;
  int at[kMaxStages];
  int buckets[kMaxStages];
};

// per-cloud record of the multi-launch form (below): grid geometry, voxel count, crowded-voxel counter, per-block counts
constexpr int kMultiMaxBlocks = 256;   // 1024-point blocks per cloud
struct GsMeta {
  float org[3];
  int M;
  unsigned long long nx, ny;
  int nbig, pad;
  int blk[kMultiMaxBlocks];
};

struct GridArgs {
  GsMeta* meta;          // [batch] (multi-launch form only)
  const float* points;
  const int64_t* lengths;
  int batch;
  float voxel;
  float* tmp_points;     // [n_points,3] per-cloud results at the cloud's input offset
  int64_t* out_lengths;  // [batch]
  // scratch, carved per cloud from arrays sized for the whole batch
  unsigned long long* ht_keys;  // 4*n + 64*batch
  unsigned* ht_first;           // same
  unsigned* ht_rank;            // same
  unsigned* pt_slot;            // n
  int* scan;                    // n
  unsigned long long* ekey;     // n
  int* ecnt;                    // n
  int* ebase;                   // n
  int* efill;                   // n
  int* list;                    // n
  float* epts;                  // 3n
  int* order_a;                 // n
  int* order_b;                 // n
  int* bt;                      // n
  int* off;                     // n
  int* tmp;                     // n
  int* bf;                      // 3*n + 64*batch  (bucket first-time)
  int* bc;                      // same             (bucket count)
  int* bl;                      // same             (bucket fill)
  Schedule sched;
};

__device__ __forceinline__ int block_prefix(int v, int* lds, int& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  if (w == 0) {
    int x = lane < (kT / 64) ? lds[lane] : 0;
    int s = x;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      int t = __shfl_up(s, o, 64);
      if (lane >= o) s += t;
    }
    if (lane < kT / 64) lds[lane] = s - x;
    if (lane == kT / 64 - 1) lds[kT / 64] = s;
  }
  __syncthreads();
  int res = inc - v + lds[w];
  total = lds[kT / 64];
  __syncthreads();
  return res;
}

// Exclusive scan of get(i), i in [0,n), written to out[i]; `reverse` scans from the high end
// (suffix sums).  Returns the total.  All threads of the block must call it.
template <typename F>
__device__ int block_scan(int n, F get, int* out, int* lds, bool reverse) {
  const int per = (n + kT - 1) / kT;
  int c0 = threadIdx.x * per;
  if (c0 > n) c0 = n;
  int c1 = c0 + per;
  if (c1 > n) c1 = n;
  int local = 0;
  for (int i = c0; i < c1; ++i) local += get(reverse ? n - 1 - i : i);
  int total;
  int pre = block_prefix(local, lds, total);
  for (int i = c0; i < c1; ++i) {
    const int j = reverse ? n - 1 - i : i;
    const int v = get(j);
    out[j] = pre;
    pre += v;
  }
  __syncthreads();
  return total;
}

// P7 with everything but the voxel keys in LDS.  Thread `tid` owns the stage elements t in [c0, c1) through all
// phases of a stage, so bucket ids, firstness and run offsets stay in registers.  bf: first time of a bucket,
// later its run offset; bcl: bucket count (low 16 bits) and fill cursor (high 16 bits).  Returns the final list.
__device__ unsigned short* replay_lds(const GridArgs& a, const unsigned long long* __restrict__ ekey, int M,
                                      unsigned short* cur, unsigned short* nxt, unsigned short* tmp, unsigned* bf,
                                      unsigned* bcl, int* s_scan) {
  const int tid = threadIdx.x;
  for (int j = 0; j < a.sched.n; ++j) {
    const int k0 = a.sched.at[j];
    if (k0 >= M) break;
    int k1 = (j + 1 < a.sched.n) ? a.sched.at[j + 1] : 0x7fffffff;
    if (k1 > M) k1 = M;
    const unsigned B = static_cast<unsigned>(a.sched.buckets[j]);
    const int n = k1;
    for (unsigned x = tid; x < B; x += kT) {
      bf[x] = 0xffffffffu;
      bcl[x] = 0;
    }
    const int per = (n + kT - 1) / kT;
    const int c0 = min(tid * per, n), c1 = min(c0 + per, n);
    unsigned long long key[kOwn];
#pragma unroll
    for (int u = 0; u < kOwn; ++u) {  // all key loads of the thread in flight together
      const int t = c0 + u;
      key[u] = 0;
      if (t < c1) key[u] = ekey[t < k0 ? cur[t] : t];
    }
    __syncthreads();
    unsigned bk[kOwn];
#pragma unroll
    for (int u = 0; u < kOwn; ++u) {
      const int t = c0 + u;
      bk[u] = 0;
      if (t < c1) {
        bk[u] = static_cast<unsigned>(key[u] % static_cast<unsigned long long>(B));
        atomicMin(&bf[bk[u]], static_cast<unsigned>(t));
        atomicAdd(&bcl[bk[u]], 1u);
      }
    }
    __syncthreads();
    // run offsets: buckets in descending first time = exclusive suffix sums of (first ? count : 0)
    int val[kOwn], local = 0;
    unsigned first_mask = 0;
#pragma unroll
    for (int u = 0; u < kOwn; ++u) {
      const int t = c0 + u;
      val[u] = 0;
      if (t < c1 && bf[bk[u]] == static_cast<unsigned>(t)) {
        first_mask |= 1u << u;
        val[u] = static_cast<int>(bcl[bk[u]] & 0xffffu);
      }
      local += val[u];
    }
    int total;
    const int pre = block_prefix(local, s_scan, total);  // (barriers inside: every firstness test is done)
    int running = total - pre - local;
    int offv[kOwn];
#pragma unroll
    for (int u = kOwn - 1; u >= 0; --u) {
      offv[u] = running;
      running += val[u];
    }
#pragma unroll
    for (int u = 0; u < kOwn; ++u)
      if (first_mask & (1u << u)) bf[bk[u]] = static_cast<unsigned>(offv[u]);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kOwn; ++u) {
      const int t = c0 + u;
      if (t < c1) {
        const unsigned slot = atomicAdd(&bcl[bk[u]], 0x10000u) >> 16;
        tmp[bf[bk[u]] + slot] = static_cast<unsigned short>(t);
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kOwn; ++u)
      if (first_mask & (1u << u)) {
        const int c = static_cast<int>(bcl[bk[u]] & 0xffffu);
        unsigned short* L = tmp + offv[u];
        for (int x = 1; x < c; ++x) {  // newest first inside a bucket
          const unsigned short v = L[x];
          int y = x - 1;
          while (y >= 0 && L[y] < v) {
            L[y + 1] = L[y];
            --y;
          }
          L[y + 1] = v;
        }
      }
    __syncthreads();
    for (int pos = c0; pos < c1; ++pos) {
      const int t = tmp[pos];
      nxt[pos] = t < k0 ? cur[t] : static_cast<unsigned short>(t);
    }
    __syncthreads();
    unsigned short* sw = cur;
    cur = nxt;
    nxt = sw;
  }
  return cur;
}

// One crowded voxel (more than 8 points) by one wavefront: rank sort of the point indices (lane = list entry), then the sums
// in ascending point order from registers (the order fixes the fp32 result).  sorted64: 64 ints of LDS private to the wavefront.
__device__ __forceinline__ void crowded_voxel(const float* P, const int* list, int* cur, const int* ecnt, const int* ebase,
                                              float* epts, int e, int lane, int* sorted64) {
  const int c = ld_agent(&ecnt[e]), base = ebase[e];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  if (c <= 64) {
    const int mine = lane < c ? list[base + lane] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < c; ++j) rank += __builtin_amdgcn_readlane(mine, j) < mine ? 1 : 0;
    if (lane < c) sorted64[rank] = mine;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int i = lane < c ? sorted64[lane] : 0;
    const float px = P[3 * i], py = P[3 * i + 1], pz = P[3 * i + 2];
    for (int j = 0; j < c; ++j) {
      sx += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px), j));
      sy += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py), j));
      sz += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz), j));
    }
  } else {
    // more than a wavefront of points in one voxel (a degenerate cloud can put ALL its points there): the same
    // rank sort in chunks of 64 against chunks of 64 -- c^2/64 wavefront steps instead of one thread's c^2/4
    // memory round trips -- with the sorted indices in `cur` (a P7 array, free until then)
    for (int m0 = 0; m0 < c; m0 += 64) {
      const int mine = m0 + lane < c ? list[base + m0 + lane] : 0x7fffffff;
      int rank = 0;
      for (int o0 = 0; o0 < c; o0 += 64) {
        const int other = o0 + lane < c ? list[base + o0 + lane] : 0x7fffffff;
        const int cnt = min(64, c - o0);
        for (int j = 0; j < cnt; ++j) rank += __builtin_amdgcn_readlane(other, j) < mine ? 1 : 0;
      }
      if (m0 + lane < c) cur[base + rank] = mine;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // the sorted list was written by other lanes
    for (int m0 = 0; m0 < c; m0 += 64) {
      const int i = m0 + lane < c ? ld_agent(&cur[base + m0 + lane]) : 0;
      const float px = P[3 * i], py = P[3 * i + 1], pz = P[3 * i + 2];
      const int cnt = min(64, c - m0);
      for (int j = 0; j < cnt; ++j) {
        sx += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px), j));
        sy += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py), j));
        sz += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz), j));
      }
    }
  }
  if (lane == 0) {
    const float wgt = static_cast<float>(1.0 / static_cast<double>(c));
    epts[3 * e] = sx * wgt;
    epts[3 * e + 1] = sy * wgt;
    epts[3 * e + 2] = sz * wgt;
  }
}

constexpr int kMaskChunks = 64;  // 64-point chunks per wavefront whose ballots P2 keeps in LDS
constexpr int kBigCap = 2048;  // work list of crowded voxels; beyond it the owning thread sorts serially
#ifdef RDM_GS_TIMING
__device__ unsigned long long rdm_gs_clk[16];  // tools/gs_phase_lab.hip: shader-clock stamps of workgroup 0, thread 0
#define GS_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) rdm_gs_clk[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GS_STAMP(k) do { } while (0)
#endif
// P7 + P8 of one cloud by one 1024-thread workgroup: replay of the hash map's growth stages (in LDS when the cloud fits,
// through the per-cloud global arrays otherwise), then the voxel barycentres in the container's iteration order
// (grid_subsampling_cpu.cpp:44-47).  s_dyn: the kLdsBytes of dynamic LDS; s_scan: kT/64 + 2 ints.
__device__ void replay_and_emit(const GridArgs& a, int64_t start, int M, const unsigned long long* ekey, const float* epts, int* cur,
                                int* nxt, int* bt, int* off, int* tmp, int* bf, int* bc, int* bl, unsigned char* s_dyn,
                                int* s_scan) {
  const int tid = threadIdx.x;
  // ---- P7: replay the container's growth stages to obtain its iteration order
  int last_b = 0;
  for (int j = 0; j < a.sched.n && a.sched.at[j] < M; ++j) last_b = a.sched.buckets[j];
  const bool in_lds = M <= kLdsM && last_b <= kLdsB;  // uniform over the block
  const unsigned short* lds_order = nullptr;
  if (in_lds) {
    unsigned short* l_cur = reinterpret_cast<unsigned short*>(s_dyn);
    unsigned short* l_nxt = l_cur + kLdsM;
    unsigned short* l_tmp = l_nxt + kLdsM;
    unsigned* l_bf = reinterpret_cast<unsigned*>(l_tmp + kLdsM);
    unsigned* l_bcl = l_bf + kLdsB;
    lds_order = replay_lds(a, ekey, M, l_cur, l_nxt, l_tmp, l_bf, l_bcl, s_scan);
  } else {
    for (int j = 0; j < a.sched.n; ++j) {
      const int k0 = a.sched.at[j];
      if (k0 >= M) break;
      int k1 = (j + 1 < a.sched.n) ? a.sched.at[j + 1] : 0x7fffffff;
      if (k1 > M) k1 = M;
      const int B = a.sched.buckets[j];
      const int n = k1;  // every element ranked < k0 is already in `cur`
      for (int x = tid; x < B; x += kT) {
        bf[x] = 0x7fffffff;
        bc[x] = 0;
        bl[x] = 0;
      }
      __syncthreads();
      for (int t = tid; t < n; t += kT) {
        const int e = t < k0 ? cur[t] : t;
        const int bkt = static_cast<int>(ekey[e] % static_cast<unsigned long long>(B));
        bt[t] = bkt;
        atomicMin(&bf[bkt], t);
        atomicAdd(&bc[bkt], 1);
      }
      __syncthreads();
      block_scan(
          n,
          [&](int t) {
            const int bkt = bt[t];
            return ld_agent(&bf[bkt]) == t ? ld_agent(&bc[bkt]) : 0;
          },
          off, s_scan, true);
      for (int t = tid; t < n; t += kT) {
        const int bkt = bt[t];
        const int base = off[ld_agent(&bf[bkt])];
        const int slot = atomicAdd(&bl[bkt], 1);
        tmp[base + slot] = t;
      }
      __syncthreads();
      for (int t = tid; t < n; t += kT) {
        const int bkt = bt[t];
        if (ld_agent(&bf[bkt]) != t) continue;
        const int c = ld_agent(&bc[bkt]);
        int* L = tmp + off[t];
        for (int x = 1; x < c; ++x) {  // newest first inside a bucket
          const int val = L[x];
          int y = x - 1;
          while (y >= 0 && L[y] < val) {
            L[y + 1] = L[y];
            --y;
          }
          L[y + 1] = val;
        }
      }
      __syncthreads();
      for (int pos = tid; pos < n; pos += kT) {
        const int t = tmp[pos];
        nxt[pos] = t < k0 ? cur[t] : t;
      }
      __syncthreads();
      int* sw = cur;
      cur = nxt;
      nxt = sw;
    }
  }

  GS_STAMP(5);
  // ---- P8: emit in list order (grid_subsampling_cpu.cpp:44-47)
  float* out = a.tmp_points + 3 * start;
  for (int pos = tid; pos < M; pos += kT) {
    const int e = in_lds ? static_cast<int>(lds_order[pos]) : cur[pos];
    out[3 * pos] = epts[3 * e];
    out[3 * pos + 1] = epts[3 * e + 1];
    out[3 * pos + 2] = epts[3 * e + 2];
  }
}

__global__ __launch_bounds__(kT) void grid_subsample_kernel(GridArgs a) {
  __shared__ int s_scan[kT / 64 + 2];
  // scratch tables of P2 and P6 live in the dynamic LDS that P7's replay uses afterwards (144 KB, always allocated)
  extern __shared__ __align__(16) unsigned char s_dyn[];
  static_assert(kLdsBytes >= (kT / 64) * kMaskChunks * 8 + kBigCap * 4 + (kT / 64) * 64 * 4, "dynamic LDS too small");
  unsigned long long (*s_mask)[kMaskChunks] = reinterpret_cast<unsigned long long (*)[kMaskChunks]>(s_dyn);  // P2 ballots
  int* s_big = reinterpret_cast<int*>(s_dyn + (kT / 64) * kMaskChunks * 8);  // P6: voxels with more than 8 points
  int (*s_sorted)[64] = reinterpret_cast<int (*)[64]>(s_dyn + (kT / 64) * kMaskChunks * 8 + kBigCap * 4);
  __shared__ int s_nbig;
  __shared__ float s_red[2 * 3 * (kT / 64)];
  __shared__ float s_org[3];
  __shared__ unsigned long long s_nxy[2];

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  int64_t start = 0;
  for (int i = 0; i < b; ++i) start += a.lengths[i];
  const int N = static_cast<int>(a.lengths[b]);
  if (N <= 0) {
    if (tid == 0) a.out_lengths[b] = 0;
    return;
  }
  const float* P = a.points + 3 * start;
  const float v = a.voxel;

  // per-cloud regions
  unsigned ht_cap = 64;
  while (ht_cap < 2u * static_cast<unsigned>(N)) ht_cap <<= 1;
  const unsigned ht_mask = ht_cap - 1;
  const size_t ht_off = 4 * static_cast<size_t>(start) + 64 * static_cast<size_t>(b);
  unsigned long long* ht_keys = a.ht_keys + ht_off;
  unsigned* ht_first = a.ht_first + ht_off;
  unsigned* ht_rank = a.ht_rank + ht_off;
  unsigned* pt_slot = a.pt_slot + start;
  unsigned long long* ekey = a.ekey + start;
  int* ecnt = a.ecnt + start;
  int* ebase = a.ebase + start;
  int* efill = a.efill + start;
  int* list = a.list + start;
  float* epts = a.epts + 3 * start;
  int* cur = a.order_a + start;
  int* nxt = a.order_b + start;
  int* bt = a.bt + start;
  int* off = a.off + start;
  int* tmp = a.tmp + start;
  const size_t b_off = 3 * static_cast<size_t>(start) + 64 * static_cast<size_t>(b);
  int* bf = a.bf + b_off;
  int* bc = a.bc + b_off;
  int* bl = a.bl + b_off;

  GS_STAMP(0);
  // ---- P0: bounding box (cloud.cpp:4-38), origin and nX, nY (grid_subsampling_cpu.cpp:9-20)
  {
    float lo[3] = {P[0], P[1], P[2]}, hi[3] = {P[0], P[1], P[2]};
    for (int i = tid; i < N; i += kT)
      for (int d = 0; d < 3; ++d) {
        const float x = P[3 * i + d];
        if (x < lo[d]) lo[d] = x;
        if (x > hi[d]) hi[d] = x;
      }
    for (int d = 0; d < 3; ++d)
      for (int o = 32; o > 0; o >>= 1) {
        lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64));
        hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64));
      }
    const int w = tid >> 6;
    if ((tid & 63) == 0)
      for (int d = 0; d < 3; ++d) {
        s_red[w * 6 + d] = lo[d];
        s_red[w * 6 + 3 + d] = hi[d];
      }
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < kT / 64; ++k)
        for (int d = 0; d < 3; ++d) {
          lo[d] = fminf(lo[d], s_red[k * 6 + d]);
          hi[d] = fmaxf(hi[d], s_red[k * 6 + 3 + d]);
        }
      const float inv = static_cast<float>(1.0 / static_cast<double>(v));
      for (int d = 0; d < 3; ++d) s_org[d] = floorf(lo[d] * inv) * v;
      const double fx = floor(static_cast<double>((hi[0] - s_org[0]) / v)) + 1.0;
      const double fy = floor(static_cast<double>((hi[1] - s_org[1]) / v)) + 1.0;
      s_nxy[0] = static_cast<unsigned long long>(static_cast<long long>(fx));
      s_nxy[1] = static_cast<unsigned long long>(static_cast<long long>(fy));
    }
    for (unsigned x = tid; x < ht_cap; x += kT) {
      ht_keys[x] = kEmpty;
      ht_first[x] = 0xFFFFFFFFu;
    }
    for (int i = tid; i < N; i += kT) {
      ecnt[i] = 0;
      efill[i] = 0;
    }
    __syncthreads();
  }
  const float ox = s_org[0], oy = s_org[1], oz = s_org[2];
  const unsigned long long nx = s_nxy[0], ny = s_nxy[1];

  // The phases below run on ONE workgroup and are bound by the latency of dependent global accesses, so every
  // loop handles a small group of points/voxels at a time with all of the group's loads (or returning atomics)
  // issued before the first result is used.
  constexpr int G = 4;

  GS_STAMP(1);
  // ---- P1: voxel key per point (grid_subsampling_cpu.cpp:28-35) + de-duplication
  for (int i0 = tid; i0 < N; i0 += G * kT) {
    unsigned long long key[G];
    unsigned slot[G];
    bool open[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int i = i0 + u * kT;
      open[u] = i < N;
      key[u] = 0;
      slot[u] = 0;
      if (open[u]) {
        const float px = P[3 * i], py = P[3 * i + 1], pz = P[3 * i + 2];
        const unsigned long long ix = static_cast<unsigned long long>(static_cast<long long>(floorf((px - ox) / v)));
        const unsigned long long iy = static_cast<unsigned long long>(static_cast<long long>(floorf((py - oy) / v)));
        const unsigned long long iz = static_cast<unsigned long long>(static_cast<long long>(floorf((pz - oz) / v)));
        key[u] = ix + nx * iy + nx * ny * iz;
        slot[u] = static_cast<unsigned>((key[u] * 0x9E3779B97F4A7C15ull) >> 40) & ht_mask;
      }
    }
    bool any = true;
    while (any) {  // linear probing, the group's CAS attempts in flight together
      unsigned long long prev[G];
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (open[u]) prev[u] = atomicCAS(&ht_keys[slot[u]], kEmpty, key[u]);
      any = false;
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (open[u]) {
          if (prev[u] == kEmpty || prev[u] == key[u]) {
            open[u] = false;
            const int i = i0 + u * kT;
            atomicMin(&ht_first[slot[u]], static_cast<unsigned>(i));
            pt_slot[i] = slot[u];
          } else {
            slot[u] = (slot[u] + 1) & ht_mask;
            any = true;
          }
        }
    }
  }
  __syncthreads();

  GS_STAMP(2);
  // ---- P2: rank distinct keys by first occurrence (= insertion order into the reference's map).  Wavefront w owns
  // the contiguous points [w0, w1) and walks them 64 at a time, lane = point: the slot reads are coalesced, the
  // "first" flags of a chunk are a ballot (kept in LDS for the second sweep when the cloud has at most 64 k points,
  // recomputed otherwise), ranks are a running count + the lane's population count, so the key list is written in order.
  // (One CU resolves about one scattered address per clock -- tools/gs_phase_lab.hip -- and with a thread owning
  // CONTIGUOUS points every one of these accesses was scattered.)
  int M;
  {
    const int lane = tid & 63, w = tid >> 6;
    constexpr int NW = kT / 64;
    const int seg = ((N + NW * 64 - 1) / (NW * 64)) * 64;
    const int w0 = min(w * seg, N), w1 = min(w0 + seg, N);
    const bool keep = seg / 64 <= kMaskChunks;
    auto first_masks = [&](int base, unsigned (&sl)[G], unsigned long long (&mk)[G]) {
#pragma unroll
      for (int u = 0; u < G; ++u) sl[u] = base + 64 * u + lane < w1 ? pt_slot[base + 64 * u + lane] : 0u;
      unsigned f[G];
#pragma unroll
      for (int u = 0; u < G; ++u) f[u] = base + 64 * u + lane < w1 ? ld_agent(&ht_first[sl[u]]) : 0xffffffffu;
#pragma unroll
      for (int u = 0; u < G; ++u) mk[u] = __ballot(f[u] == static_cast<unsigned>(base + 64 * u + lane));
    };
    int local = 0;  // wavefront-uniform
    for (int base = w0; base < w1; base += 64 * G) {
      unsigned sl[G];
      unsigned long long mk[G];
      first_masks(base, sl, mk);
#pragma unroll
      for (int u = 0; u < G; ++u) {
        local += __popcll(mk[u]);
        if (keep && lane == 0 && base + 64 * u < w1) s_mask[w][(base - w0) / 64 + u] = mk[u];
      }
    }
    int total;
    int rank = block_prefix(lane == 63 ? local : 0, s_scan, total);  // lane 63: the count of all earlier wavefronts
    rank = __shfl(rank, 63, 64);
    M = total;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = w0; base < w1; base += 64 * G) {
      unsigned sl[G];
      unsigned long long mk[G];
      if (keep) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          mk[u] = base + 64 * u < w1 ? s_mask[w][(base - w0) / 64 + u] : 0ull;
          sl[u] = ((mk[u] >> lane) & 1ull) ? pt_slot[base + 64 * u + lane] : 0u;
        }
      } else {
        first_masks(base, sl, mk);
      }
      unsigned long long kk[G];
#pragma unroll
      for (int u = 0; u < G; ++u) kk[u] = ((mk[u] >> lane) & 1ull) ? ld_agent(&ht_keys[sl[u]]) : 0ull;
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if ((mk[u] >> lane) & 1ull) {
          const int r = rank + __popcll(mk[u] & below);
          ht_rank[sl[u]] = static_cast<unsigned>(r);
          ekey[r] = kk[u];
        }
        rank += __popcll(mk[u]);
      }
    }
  }
  __syncthreads();

  GS_STAMP(3);
  // ---- P3..P6: per-voxel point lists in ascending point order, sequential fp32 sums
  // (16 points / ~10 voxels per thread at the first level: these loops take 8 points or 2 voxels per trip -- every
  // trip is a chain of 3-4 dependent memory round trips of ~1 us with nothing else on the CU to hide them)
  constexpr int G8 = 8;
  int* pt_rank = tmp;  // voxel rank of every point (tmp is a P7 array, free until then)
  for (int i0 = tid; i0 < N; i0 += G8 * kT) {
    unsigned sl[G8], rk[G8];
#pragma unroll
    for (int u = 0; u < G8; ++u) sl[u] = i0 + u * kT < N ? pt_slot[i0 + u * kT] : 0u;
#pragma unroll
    for (int u = 0; u < G8; ++u) rk[u] = i0 + u * kT < N ? ht_rank[sl[u]] : 0u;
#pragma unroll
    for (int u = 0; u < G8; ++u)
      if (i0 + u * kT < N) {
        atomicAdd(&ecnt[rk[u]], 1);
        pt_rank[i0 + u * kT] = static_cast<int>(rk[u]);
      }
  }
  __syncthreads();
  GS_STAMP(7);
  block_scan(M, [&](int e) { return ld_agent(&ecnt[e]); }, ebase, s_scan, false);
  GS_STAMP(8);
  // the fill cursor of a voxel starts at its list base: one returning atomic per point gives the list position
  for (int e = tid; e < M; e += kT) efill[e] = ebase[e];
  __syncthreads();
  for (int i0 = tid; i0 < N; i0 += G8 * kT) {
    int rk[G8], pos[G8];
#pragma unroll
    for (int u = 0; u < G8; ++u) rk[u] = i0 + u * kT < N ? pt_rank[i0 + u * kT] : 0;
#pragma unroll
    for (int u = 0; u < G8; ++u) pos[u] = i0 + u * kT < N ? atomicAdd(&efill[rk[u]], 1) : 0;
#pragma unroll
    for (int u = 0; u < G8; ++u)
      if (i0 + u * kT < N) list[pos[u]] = i0 + u * kT;
  }
  __syncthreads();
  constexpr int V = 2;  // voxels per trip
  if (tid == 0) s_nbig = 0;
  __syncthreads();
  GS_STAMP(9);
  for (int e0 = tid; e0 < M; e0 += V * kT) {
    int cnt[V], base[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int e = e0 + q * kT;
      cnt[q] = e < M ? ld_agent(&ecnt[e]) : 0;
      base[q] = e < M ? ebase[e] : 0;
    }
    // the usual voxel holds a handful of points: sort the indices in registers (bubble network), fetch all the
    // points, then add them in ascending point order (SampledData::update, grid_subsampling_cpu.h:17-20)
    int id[V][8];
#pragma unroll
    for (int q = 0; q < V; ++q)
#pragma unroll
      for (int x = 0; x < 8; ++x) id[q][x] = (cnt[q] <= 8 && x < cnt[q]) ? list[base[q] + x] : 0x7fffffff;
#pragma unroll
    for (int q = 0; q < V; ++q)
#pragma unroll
      for (int pass = 0; pass < 7; ++pass)
#pragma unroll
        for (int x = 0; x < 7 - pass; ++x) {
          const int lo = min(id[q][x], id[q][x + 1]), hi = max(id[q][x], id[q][x + 1]);
          id[q][x] = lo;
          id[q][x + 1] = hi;
        }
    float px[V][8], py[V][8], pz[V][8];
#pragma unroll
    for (int q = 0; q < V; ++q)
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int i = (cnt[q] <= 8 && x < cnt[q]) ? id[q][x] : 0;
        px[q][x] = P[3 * i];
        py[q][x] = P[3 * i + 1];
        pz[q][x] = P[3 * i + 2];
      }
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int e = e0 + q * kT, c = cnt[q];
      if (e >= M) continue;
      float sx = 0.f, sy = 0.f, sz = 0.f;
      if (c <= 8) {
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (x < c) {
            sx += px[q][x];
            sy += py[q][x];
            sz += pz[q][x];
          }
      } else {
        // a crowded voxel (near the sensor): a whole wavefront takes it below; the serial form -- one thread, an
        // insertion sort with a memory round trip per step -- was most of this phase's time
        const int slot = atomicAdd(&s_nbig, 1);  // (a full work list leaves the voxel to this thread, serially)
        if (slot < kBigCap) {
          s_big[slot] = e;
          continue;
        }
        int* L = list + base[q];
        for (int x = 1; x < c; ++x) {  // insertion sort
          const int val = L[x];
          int y = x - 1;
          while (y >= 0 && L[y] > val) {
            L[y + 1] = L[y];
            --y;
          }
          L[y + 1] = val;
        }
        for (int x = 0; x < c; ++x) {
          const int i = L[x];
          sx += P[3 * i];
          sy += P[3 * i + 1];
          sz += P[3 * i + 2];
        }
      }
      const float wgt = static_cast<float>(1.0 / static_cast<double>(c));
      epts[3 * e] = sx * wgt;
      epts[3 * e + 1] = sy * wgt;
      epts[3 * e + 2] = sz * wgt;
    }
  }
  __syncthreads();
  GS_STAMP(10);
  {  // crowded voxels, one wavefront each: rank sort of the point indices (lane = list entry), then the sums in
     // ascending point order from registers (the order fixes the fp32 result)
    const int lane = tid & 63, w = tid >> 6;
    const int nbig = min(s_nbig, kBigCap);
    for (int k = w; k < nbig; k += kT / 64) {
      crowded_voxel(P, list, cur, ecnt, ebase, epts, s_big[k], lane, s_sorted[w]);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();

  GS_STAMP(4);
  // ---- P7 + P8: replay the container's growth stages to obtain its iteration order, emit in that order
  replay_and_emit(a, start, M, ekey, epts, cur, nxt, bt, off, tmp, bf, bc, bl, s_dyn, s_scan);
  if (tid == 0) a.out_lengths[b] = M;
  GS_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-launch form for large clouds (the first pyramid level: 16-20 k points per cloud).  The single-workgroup kernel
// above spends three quarters of such a call in phases P0-P6 -- keys, de-duplication, first-occurrence ranks, per-voxel
// lists and sums -- all of them bound by ONE CU's rate for scattered accesses.  Here each phase is its own launch over
// as many workgroups as the cloud has 1024-point blocks (kernel boundaries are the grid-wide barriers), and only the
// inherently ordered tail -- the hash-map order replay and the emit, replay_and_emit() -- stays on one workgroup per
// cloud.  Same arithmetic, same tables, same output bit for bit (tests/test_native_gpu.py compares both forms).
struct CloudView {
  int64_t start;
  int N;
  const float* P;
  unsigned ht_mask, ht_cap;
  unsigned long long* ht_keys;
  unsigned* ht_first;
  unsigned* ht_rank;
  unsigned* pt_slot;
  unsigned long long* ekey;
  int *ecnt, *ebase, *efill, *list, *cur, *nxt, *bt, *off, *tmp, *bf, *bc, *bl;
  float* epts;
};
__device__ __forceinline__ CloudView cloud_view(const GridArgs& a, int b) {
  CloudView v;
  int64_t start = 0;
  for (int i = 0; i < b; ++i) start += a.lengths[i];
  v.start = start;
  v.N = static_cast<int>(a.lengths[b]);
  v.P = a.points + 3 * start;
  unsigned cap = 64;
  while (cap < 2u * static_cast<unsigned>(v.N > 0 ? v.N : 0)) cap <<= 1;
  v.ht_cap = cap;
  v.ht_mask = cap - 1;
  const size_t ht_off = 4 * static_cast<size_t>(start) + 64 * static_cast<size_t>(b);
  v.ht_keys = a.ht_keys + ht_off;
  v.ht_first = a.ht_first + ht_off;
  v.ht_rank = a.ht_rank + ht_off;
  v.pt_slot = a.pt_slot + start;
  v.ekey = a.ekey + start;
  v.ecnt = a.ecnt + start;
  v.ebase = a.ebase + start;
  v.efill = a.efill + start;
  v.list = a.list + start;
  v.epts = a.epts + 3 * start;
  v.cur = a.order_a + start;
  v.nxt = a.order_b + start;
  v.bt = a.bt + start;
  v.off = a.off + start;
  v.tmp = a.tmp + start;
  const size_t b_off = 3 * static_cast<size_t>(start) + 64 * static_cast<size_t>(b);
  v.bf = a.bf + b_off;
  v.bc = a.bc + b_off;
  v.bl = a.bl + b_off;
  return v;
}

// M0: bounding box, origin, nX, nY (P0 of the single-workgroup kernel) and the empty tables; one workgroup per cloud
__global__ __launch_bounds__(kT) void gs_prepare_kernel(GridArgs a) {
  __shared__ float s_red[2 * 3 * (kT / 64)];
  const int b = blockIdx.x, tid = threadIdx.x;
  const CloudView c = cloud_view(a, b);
  GsMeta* meta = a.meta + b;
  if (c.N <= 0) {
    if (tid == 0) {
      a.out_lengths[b] = 0;
      meta->M = 0;
      meta->nbig = 0;
    }
    return;
  }
  const float* P = c.P;
  const float v = a.voxel;
  float lo[3] = {P[0], P[1], P[2]}, hi[3] = {P[0], P[1], P[2]};
  for (int i = tid; i < c.N; i += kT)
    for (int d = 0; d < 3; ++d) {
      const float x = P[3 * i + d];
      if (x < lo[d]) lo[d] = x;
      if (x > hi[d]) hi[d] = x;
    }
  for (int d = 0; d < 3; ++d)
    for (int o = 32; o > 0; o >>= 1) {
      lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64));
      hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64));
    }
  const int w = tid >> 6;
  if ((tid & 63) == 0)
    for (int d = 0; d < 3; ++d) {
      s_red[w * 6 + d] = lo[d];
      s_red[w * 6 + 3 + d] = hi[d];
    }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < kT / 64; ++k)
      for (int d = 0; d < 3; ++d) {
        lo[d] = fminf(lo[d], s_red[k * 6 + d]);
        hi[d] = fmaxf(hi[d], s_red[k * 6 + 3 + d]);
      }
    const float inv = static_cast<float>(1.0 / static_cast<double>(v));
    float org[3];
    for (int d = 0; d < 3; ++d) org[d] = floorf(lo[d] * inv) * v;
    const double fx = floor(static_cast<double>((hi[0] - org[0]) / v)) + 1.0;
    const double fy = floor(static_cast<double>((hi[1] - org[1]) / v)) + 1.0;
    for (int d = 0; d < 3; ++d) meta->org[d] = org[d];
    meta->nx = static_cast<unsigned long long>(static_cast<long long>(fx));
    meta->ny = static_cast<unsigned long long>(static_cast<long long>(fy));
    meta->M = 0;
    meta->nbig = 0;
  }
  for (unsigned x = tid; x < c.ht_cap; x += kT) {
    c.ht_keys[x] = kEmpty;
    c.ht_first[x] = 0xFFFFFFFFu;
  }
  for (int i = tid; i < c.N; i += kT) c.ecnt[i] = 0;
}

// M1: voxel key per point + de-duplication (P1); grid = (1024-point blocks, clouds)
__global__ __launch_bounds__(kT) void gs_keys_kernel(GridArgs a) {
  const CloudView c = cloud_view(a, blockIdx.y);
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= c.N) return;
  const GsMeta* meta = a.meta + blockIdx.y;
  const float v = a.voxel, ox = meta->org[0], oy = meta->org[1], oz = meta->org[2];
  const unsigned long long nx = meta->nx, ny = meta->ny;
  const float px = c.P[3 * i], py = c.P[3 * i + 1], pz = c.P[3 * i + 2];
  const unsigned long long ix = static_cast<unsigned long long>(static_cast<long long>(floorf((px - ox) / v)));
  const unsigned long long iy = static_cast<unsigned long long>(static_cast<long long>(floorf((py - oy) / v)));
  const unsigned long long iz = static_cast<unsigned long long>(static_cast<long long>(floorf((pz - oz) / v)));
  const unsigned long long key = ix + nx * iy + nx * ny * iz;
  unsigned slot = static_cast<unsigned>((key * 0x9E3779B97F4A7C15ull) >> 40) & c.ht_mask;
  for (;;) {  // linear probing
    const unsigned long long prev = atomicCAS(&c.ht_keys[slot], kEmpty, key);
    if (prev == kEmpty || prev == key) break;
    slot = (slot + 1) & c.ht_mask;
  }
  atomicMin(&c.ht_first[slot], static_cast<unsigned>(i));
  c.pt_slot[i] = slot;
}

// M2: how many points of this block are the first of their voxel
__global__ __launch_bounds__(kT) void gs_first_count_kernel(GridArgs a) {
  __shared__ int s_cnt[kT / 64];
  const CloudView c = cloud_view(a, blockIdx.y);
  if (static_cast<int>(blockIdx.x) * kT >= c.N) return;  // (whole workgroup)
  const int i = blockIdx.x * kT + threadIdx.x;
  const bool first = i < c.N && c.ht_first[c.pt_slot[i]] == static_cast<unsigned>(i);
  const unsigned long long m = __ballot(first);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int k = 0; k < kT / 64; ++k) t += s_cnt[k];
    a.meta[blockIdx.y].blk[blockIdx.x] = t;
  }
}

// M3: rank of every distinct key by first occurrence (= insertion order into the reference's map) and the key list (P2)
__global__ __launch_bounds__(kT) void gs_rank_kernel(GridArgs a) {
  __shared__ int s_cnt[kT / 64 + 1];
  __shared__ int s_base;
  const CloudView c = cloud_view(a, blockIdx.y);
  if (static_cast<int>(blockIdx.x) * kT >= c.N) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const GsMeta* meta = a.meta + blockIdx.y;
  // voxels first seen in earlier blocks (at most kMultiMaxBlocks = 256 counts: one per thread of the first four wavefronts)
  int part = tid < static_cast<int>(blockIdx.x) ? meta->blk[tid] : 0;
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) s_cnt[w] = part;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int k = 0; k < kT / 64; ++k) t += s_cnt[k];
    s_base = t;
  }
  __syncthreads();
  const int base = s_base;
  __syncthreads();
  const int i = blockIdx.x * kT + tid;
  unsigned slot = 0;
  bool first = false;
  if (i < c.N) {
    slot = c.pt_slot[i];
    first = c.ht_first[slot] == static_cast<unsigned>(i);
  }
  const unsigned long long m = __ballot(first);
  if (lane == 0) s_cnt[w] = __popcll(m);
  __syncthreads();
  int before = 0;
  for (int k = 0; k < w; ++k) before += s_cnt[k];
  if (first) {
    const int r = base + before + __popcll(m & ((1ull << lane) - 1ull));
    c.ht_rank[slot] = static_cast<unsigned>(r);
    c.ekey[r] = c.ht_keys[slot];
  }
}

// M4: points per voxel, voxel rank of every point (first half of P3..P6)
__global__ __launch_bounds__(kT) void gs_count_kernel(GridArgs a) {
  const CloudView c = cloud_view(a, blockIdx.y);
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= c.N) return;
  const int rk = static_cast<int>(c.ht_rank[c.pt_slot[i]]);
  atomicAdd(&c.ecnt[rk], 1);
  c.tmp[i] = rk;
}

// M5: voxel count, list bases (exclusive scan of the counts), fill cursors; one workgroup per cloud
__global__ __launch_bounds__(kT) void gs_scan_kernel(GridArgs a) {
  __shared__ int s_scan[kT / 64 + 2];
  __shared__ int s_m;
  const CloudView c = cloud_view(a, blockIdx.x);
  if (c.N <= 0) return;
  GsMeta* meta = a.meta + blockIdx.x;
  const int nblk = (c.N + kT - 1) / kT;
  if (threadIdx.x == 0) {
    int t = 0;
    for (int k = 0; k < nblk; ++k) t += meta->blk[k];
    s_m = t;
    meta->M = t;
  }
  __syncthreads();
  const int M = s_m;
  block_scan(M, [&](int e) { return c.ecnt[e]; }, c.ebase, s_scan, false);
  for (int e = threadIdx.x; e < M; e += kT) c.efill[e] = c.ebase[e];
}

// M6: per-voxel point lists (one returning atomic per point gives the list position)
__global__ __launch_bounds__(kT) void gs_fill_kernel(GridArgs a) {
  const CloudView c = cloud_view(a, blockIdx.y);
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= c.N) return;
  const int pos = atomicAdd(&c.efill[c.tmp[i]], 1);
  c.list[pos] = i;
}

// M7: per-voxel sums in ascending point order (SampledData::update, grid_subsampling_cpu.h:17-20); one thread per voxel, the
// crowded voxels (more than 8 points) go to a work list for M8
__global__ __launch_bounds__(256) void gs_sum_kernel(GridArgs a) {
  const CloudView c = cloud_view(a, blockIdx.y);
  GsMeta* meta = a.meta + blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= meta->M) return;
  const int cnt = c.ecnt[e], base = c.ebase[e];
  if (cnt > 8) {
    c.off[atomicAdd(&meta->nbig, 1)] = e;  // (at most M entries; `off` is a replay array, free until then)
    return;
  }
  int id[8];
#pragma unroll
  for (int x = 0; x < 8; ++x) id[x] = x < cnt ? c.list[base + x] : 0x7fffffff;
#pragma unroll
  for (int pass = 0; pass < 7; ++pass)
#pragma unroll
    for (int x = 0; x < 7 - pass; ++x) {
      const int lo = min(id[x], id[x + 1]), hi = max(id[x], id[x + 1]);
      id[x] = lo;
      id[x + 1] = hi;
    }
  float px[8], py[8], pz[8];
#pragma unroll
  for (int x = 0; x < 8; ++x) {
    const int i = x < cnt ? id[x] : 0;
    px[x] = c.P[3 * i];
    py[x] = c.P[3 * i + 1];
    pz[x] = c.P[3 * i + 2];
  }
  float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
  for (int x = 0; x < 8; ++x)
    if (x < cnt) {
      sx += px[x];
      sy += py[x];
      sz += pz[x];
    }
  const float wgt = static_cast<float>(1.0 / static_cast<double>(cnt));
  c.epts[3 * e] = sx * wgt;
  c.epts[3 * e + 1] = sy * wgt;
  c.epts[3 * e + 2] = sz * wgt;
}

// M8: the crowded voxels, one wavefront each
__global__ __launch_bounds__(256) void gs_crowded_kernel(GridArgs a) {
  __shared__ int s_sorted[4][64];
  const CloudView c = cloud_view(a, blockIdx.y);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nbig = ld_agent(&a.meta[blockIdx.y].nbig);
  for (int k = blockIdx.x * 4 + w; k < nbig; k += gridDim.x * 4) {
    crowded_voxel(c.P, c.list, c.cur, c.ecnt, c.ebase, c.epts, c.off[k], lane, s_sorted[w]);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// M9: hash-map order replay + emit (P7, P8); one workgroup per cloud, the replay's tables in dynamic LDS
__global__ __launch_bounds__(kT) void gs_replay_kernel(GridArgs a) {
  __shared__ int s_scan[kT / 64 + 2];
  extern __shared__ __align__(16) unsigned char s_dyn[];
  const int b = blockIdx.x;
  const CloudView c = cloud_view(a, b);
  if (c.N <= 0) return;  // (out_lengths was written by gs_prepare_kernel)
  const int M = a.meta[b].M;
  replay_and_emit(a, c.start, M, c.ekey, c.epts, c.cur, c.nxt, c.bt, c.off, c.tmp, c.bf, c.bc, c.bl, s_dyn, s_scan);
  if (threadIdx.x == 0) a.out_lengths[b] = M;
}

// Stack the per-cloud results contiguously (grid_subsampling_cpu.cpp:67-68).
__global__ void compact_clouds_kernel(const float* tmp_points, const int64_t* in_lengths,
                                      const int64_t* out_lengths, int batch, float* out_points) {
  const int b = blockIdx.y;
  int64_t src = 0, dst = 0;
  for (int i = 0; i < b; ++i) {
    src += in_lengths[i];
    dst += out_lengths[i];
  }
  const int64_t n = 3 * out_lengths[b];
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    out_points[3 * dst + i] = tmp_points[3 * src + i];
}

int fill_schedule(int64_t max_elems, Schedule* s) {
  std::__detail::_Prime_rehash_policy pol;
  std::size_t bkt = 1;
  s->n = 0;
  // Only the insert that overflows the current threshold can trigger a rehash, so jump there.
  std::size_t k = 0;
  while (static_cast<int64_t>(k) < max_elems && s->n < kMaxStages) {
    auto r = pol._M_need_rehash(bkt, k, 1);
    if (r.first) {
      bkt = r.second;
      s->at[s->n] = static_cast<int>(k);
      s->buckets[s->n] = static_cast<int>(bkt);
      ++s->n;
      k = bkt;  // max_load_factor 1.0: the next growth happens when size reaches bucket_count
    } else {
      ++k;
    }
  }
  return s->n;
}

}  // namespace

extern "C" int rdm_rehash_schedule(int64_t max_elems, int64_t* at_host, int64_t* buckets_host,
                                   int cap) {
  std::__detail::_Prime_rehash_policy pol;
  std::size_t bkt = 1;
  int n = 0;
  for (int64_t k = 0; k < max_elems; ++k) {  // exhaustive on purpose: this is the check of fill_schedule
    auto r = pol._M_need_rehash(bkt, static_cast<std::size_t>(k), 1);
    if (r.first) {
      bkt = r.second;
      if (n < cap) {
        at_host[n] = k;
        buckets_host[n] = static_cast<int64_t>(bkt);
      }
      ++n;
    }
  }
  return n;
}

extern "C" size_t rdm_grid_subsample_workspace_bytes(int64_t n_points, int batch) {
  Arena a(nullptr, 0);
  const size_t n = static_cast<size_t>(n_points > 0 ? n_points : 1);
  const size_t ht = 4 * n + 64 * static_cast<size_t>(batch);
  const size_t bk = 3 * n + 64 * static_cast<size_t>(batch);
  a.take<float>(3 * n);
  a.take<unsigned long long>(ht);
  a.take<unsigned>(ht);
  a.take<unsigned>(ht);
  a.take<unsigned>(n);
  a.take<int>(n);
  a.take<unsigned long long>(n);
  for (int i = 0; i < 4; ++i) a.take<int>(n);
  a.take<float>(3 * n);
  for (int i = 0; i < 5; ++i) a.take<int>(n);
  for (int i = 0; i < 3; ++i) a.take<int>(bk);
  a.take<GsMeta>(static_cast<size_t>(batch > 0 ? batch : 1));
  return a.off;
}

extern "C" int rdm_grid_subsample(const float* points, int64_t n_points, const int64_t* lengths,
                                  int batch, float voxel_size, float* out_points,
                                  int64_t* out_lengths, void* ws, size_t ws_bytes, void* stream) {
  return rdm::grid_subsample_mode(points, n_points, lengths, batch, voxel_size, out_points, out_lengths, ws, ws_bytes, stream, 0);
}

extern "C" int rdm_grid_subsample_form(const float* points, int64_t n_points, const int64_t* lengths, int batch,
                                       float voxel_size, float* out_points, int64_t* out_lengths, void* ws, size_t ws_bytes,
                                       void* stream, int form) {
  RDM_REQUIRE(form >= 0 && form <= 2, "rdm_grid_subsample_form: form must be 0 (by size), 1 (one workgroup per cloud) or 2 (multi-launch)");
  return rdm::grid_subsample_mode(points, n_points, lengths, batch, voxel_size, out_points, out_lengths, ws, ws_bytes, stream, form);
}

int rdm::grid_subsample_mode(const float* points, int64_t n_points, const int64_t* lengths, int batch, float voxel_size,
                             float* out_points, int64_t* out_lengths, void* ws, size_t ws_bytes, void* stream, int mode) {
  using namespace rdm;
  RDM_REQUIRE(points && lengths && out_points && out_lengths, "rdm_grid_subsample: null pointer");
  RDM_REQUIRE(batch > 0 && n_points >= 0 && n_points < (1ll << 30),
              "rdm_grid_subsample: bad sizes (n_points=%lld batch=%d)", (long long)n_points, batch);
  RDM_REQUIRE(voxel_size > 0.f, "rdm_grid_subsample: voxel_size must be positive");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_points == 0) {
    RDM_HIP_CHECK(hipMemsetAsync(out_lengths, 0, sizeof(int64_t) * batch, st));
    return RDM_OK;
  }
  Arena ar(ws, ws_bytes);
  const size_t n = static_cast<size_t>(n_points);
  const size_t ht = 4 * n + 64 * static_cast<size_t>(batch);
  const size_t bk = 3 * n + 64 * static_cast<size_t>(batch);
  GridArgs a;
  a.points = points;
  a.lengths = lengths;
  a.batch = batch;
  a.voxel = voxel_size;
  a.out_lengths = out_lengths;
  a.tmp_points = ar.take<float>(3 * n);
  a.ht_keys = ar.take<unsigned long long>(ht);
  a.ht_first = ar.take<unsigned>(ht);
  a.ht_rank = ar.take<unsigned>(ht);
  a.pt_slot = ar.take<unsigned>(n);
  a.scan = ar.take<int>(n);
  a.ekey = ar.take<unsigned long long>(n);
  a.ecnt = ar.take<int>(n);
  a.ebase = ar.take<int>(n);
  a.efill = ar.take<int>(n);
  a.list = ar.take<int>(n);
  a.epts = ar.take<float>(3 * n);
  a.order_a = ar.take<int>(n);
  a.order_b = ar.take<int>(n);
  a.bt = ar.take<int>(n);
  a.off = ar.take<int>(n);
  a.tmp = ar.take<int>(n);
  a.bf = ar.take<int>(bk);
  a.bc = ar.take<int>(bk);
  a.bl = ar.take<int>(bk);
  a.meta = ar.take<GsMeta>(static_cast<size_t>(batch));
  if (!ar.ok) {
    set_error("rdm_grid_subsample: workspace too small (%zu < %zu bytes)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  fill_schedule(n_points + 1, &a.sched);
  // large clouds (the first pyramid level): phases P0-P6 as separate launches over many workgroups, see gs_*_kernel.
  // mode: 0 = choose by size, 1 = single-workgroup kernel, 2 = multi-launch form
  static const bool force_single = ::rdm::dev_knob("RDM_GS_SINGLE") != nullptr;  // developer knob (A/B runs)
  const bool multi = mode == 2 || (mode == 0 && !force_single && n_points >= kMultiMinPoints);
  if (multi && n_points <= static_cast<int64_t>(kMultiMaxBlocks) * kT) {
    static std::atomic<uint64_t> replay_attr{0};
    RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(gs_replay_kernel), static_cast<int>(kLdsBytes), replay_attr));
    const dim3 pts(static_cast<unsigned>(ceil_div<int64_t>(n_points, kT)), batch), vox(static_cast<unsigned>(ceil_div<int64_t>(n_points, 256)), batch);
    hipLaunchKernelGGL(gs_prepare_kernel, dim3(batch), dim3(kT), 0, st, a);
    hipLaunchKernelGGL(gs_keys_kernel, pts, dim3(kT), 0, st, a);
    hipLaunchKernelGGL(gs_first_count_kernel, pts, dim3(kT), 0, st, a);
    hipLaunchKernelGGL(gs_rank_kernel, pts, dim3(kT), 0, st, a);
    hipLaunchKernelGGL(gs_count_kernel, pts, dim3(kT), 0, st, a);
    hipLaunchKernelGGL(gs_scan_kernel, dim3(batch), dim3(kT), 0, st, a);
    hipLaunchKernelGGL(gs_fill_kernel, pts, dim3(kT), 0, st, a);
    hipLaunchKernelGGL(gs_sum_kernel, vox, dim3(256), 0, st, a);
    hipLaunchKernelGGL(gs_crowded_kernel, dim3(32, batch), dim3(256), 0, st, a);
    hipLaunchKernelGGL(gs_replay_kernel, dim3(batch), dim3(kT), kLdsBytes, st, a);
    if (int e = launch_status("grid subsample (multi-launch)")) return e;
    const int cblocks = static_cast<int>(ceil_div<int64_t>(3 * n_points, 256 * 4));
    hipLaunchKernelGGL(compact_clouds_kernel, dim3(cblocks > 0 ? cblocks : 1, batch), dim3(256), 0, st,
                       a.tmp_points, lengths, out_lengths, batch, out_points);
    return launch_status("compact_clouds_kernel");
  }
  static std::atomic<uint64_t> lds_attr{0};
  RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(grid_subsample_kernel), static_cast<int>(kLdsBytes), lds_attr));
  hipLaunchKernelGGL(grid_subsample_kernel, dim3(batch), dim3(kT), kLdsBytes, st, a);
  if (int e = launch_status("grid_subsample_kernel")) return e;
  const int blocks = static_cast<int>(ceil_div<int64_t>(3 * n_points, 256 * 4));
  hipLaunchKernelGGL(compact_clouds_kernel, dim3(blocks > 0 ? blocks : 1, batch), dim3(256), 0, st,
                     a.tmp_points, lengths, out_lengths, batch, out_points);
  return launch_status("compact_clouds_kernel");
}
