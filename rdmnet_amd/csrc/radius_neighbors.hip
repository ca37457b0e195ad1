// a2 -- fixed-radius neighbour search, bit-exact with the reference's nanoflann search.
//
// Reference: geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
// (per-cloud kd-tree, `sorted = true`), nanoflann.hpp:435-441 (metric), :249-250 (strict `<`).
// The result set of a radius query does not depend on the search structure, so the kd-tree is
// replaced by a uniform grid (cell edge >= radius) built on the fly:
//   bbox -> per-cell counts (atomics; integer sums, so the counts do not depend on their order) -> cell segments in
//   CELL ORDER (prefix over 4096-cell chunks) -> member lists -> records
//   {x,y,z,index} placed by the rank of their index inside the cell: the sorted array is a function of the points
//   alone, run to run (round 4: it also orders the queries of the KPConv tile kernel, whose GroupNorm partials follow
//   its workgroups) -> one wavefront per query scans its 27 cells with coalesced float4
//   loads, keeps hits with wave ballots, sorts (d2, index) keys bitonically in LDS.
// Float semantics (must not be contracted): d = q - s per axis, d2 = ((dx*dx)+(dy*dy))+(dz*dz),
// accept iff d2 < radius*radius (all fp32).  Ties in d2 are ordered by index (canonical order; the
// reference's std::sort leaves them unspecified).
#pragma clang fp contract(off)

#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "internal.h"

namespace {

using namespace rdm;

constexpr int kMaxCells = 1 << 20;   // cells over all clouds of one call
constexpr int kMaxBatch = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kChunkShift = 12;      // cells per chunk of the segment prefix (4096)
constexpr int kChunks = kMaxCells >> kChunkShift;  // 256

struct GridMeta {
  float org[3];
  float inv_cell;
  int dim[3];
  int cells_per_cloud;
  int total;
  int chunk_sum[kChunks];  // points per chunk of 4096 consecutive cells
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One block: bounding box of all support points, grid geometry.
// Several grids (the five levels of a pair) are built by the same launches: blockIdx.y = grid.
constexpr int kBuildMax = 8;
struct RnBuildItem {
  const float* s;
  int64_t ns;
  const int64_t* lengths;
  float radius;
  GridMeta* meta;
  int* cell_count;
  int* cell_start;
  int* pt_cell;
  int* pt_slot;
  int* cell_list;  // point indices grouped by cell (arrival order inside a cell)
  float4* sorted;
};
struct RnBuildBatch {
  RnBuildItem item[kBuildMax];
  int n, batch;
};
__global__ __launch_bounds__(1024) void rn_bbox_kernel(RnBuildBatch bb) {
  const RnBuildItem& it = bb.item[blockIdx.x];
  const float* s = it.s;
  const int64_t ns = it.ns;
  const float radius = it.radius;
  const int batch = bb.batch;
  GridMeta* meta = it.meta;
  __shared__ float red[16 * 6];
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int64_t i = threadIdx.x; i < ns; i += blockDim.x)
    for (int d = 0; d < 3; ++d) {
      const float x = s[3 * i + d];
      lo[d] = fminf(lo[d], x);
      hi[d] = fmaxf(hi[d], x);
    }
  for (int d = 0; d < 3; ++d)
    for (int o = 32; o > 0; o >>= 1) {
      lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64));
      hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64));
    }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 3; ++d) {
      red[w * 6 + d] = lo[d];
      red[w * 6 + 3 + d] = hi[d];
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k)
      for (int d = 0; d < 3; ++d) {
        lo[d] = fminf(lo[d], red[k * 6 + d]);
        hi[d] = fmaxf(hi[d], red[k * 6 + 3 + d]);
      }
    // cell edge = 1.001 * radius * m (m = smallest integer for which the grid fits kMaxCells); the
    // 0.1 % margin keeps |cell(q) - cell(s)| <= 1 for every pair with d2 < r2 despite the rounding
    // of the binning arithmetic.
    float cell = radius;
    int dim[3];
    for (int m = 1;; ++m) {
      cell = radius * 1.001f * static_cast<float>(m);
      double cells = 1.0;
      for (int d = 0; d < 3; ++d) {
        const float ext = fmaxf(hi[d] - lo[d], 0.f);
        double n = floor(static_cast<double>(ext) / static_cast<double>(cell)) + 1.0;
        if (n > 1.0e6) n = 1.0e6;
        dim[d] = static_cast<int>(n);
        cells *= n;
      }
      if (cells * batch <= static_cast<double>(kMaxCells)) break;
    }
    for (int d = 0; d < 3; ++d) {
      meta->org[d] = lo[d];
      meta->dim[d] = dim[d];
    }
    meta->inv_cell = 1.0f / cell;
    meta->cells_per_cloud = dim[0] * dim[1] * dim[2];
    meta->total = 0;
  }
  if (threadIdx.x < kChunks) meta->chunk_sum[threadIdx.x] = 0;
}

__device__ __forceinline__ void cell_of(const GridMeta& g, float x, float y, float z, int& cx,
                                        int& cy, int& cz) {
  cx = clampi(static_cast<int>(floorf((x - g.org[0]) * g.inv_cell)), 0, g.dim[0] - 1);
  cy = clampi(static_cast<int>(floorf((y - g.org[1]) * g.inv_cell)), 0, g.dim[1] - 1);
  cz = clampi(static_cast<int>(floorf((z - g.org[2]) * g.inv_cell)), 0, g.dim[2] - 1);
}

// cloud index of stacked row i (batch is tiny)
__device__ __forceinline__ int cloud_of(const int64_t* lengths, int batch, int64_t i, int64_t& begin) {
  int64_t acc = 0;
  for (int b = 0; b < batch; ++b) {
    const int64_t n = lengths[b];
    if (i < acc + n) {
      begin = acc;
      return b;
    }
    acc += n;
  }
  begin = acc;
  return batch;  // beyond the stacked rows
}

// cell_count of the cells the bounding box gave this grid (the arrays are sized for kMaxCells)
__global__ void rn_zero_kernel(RnBuildBatch bb) {
  const RnBuildItem& it = bb.item[blockIdx.y];
  const int ncell = it.meta->cells_per_cloud * bb.batch;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += gridDim.x * blockDim.x) it.cell_count[c] = 0;
}

__global__ void rn_count_kernel(RnBuildBatch bb) {
  const RnBuildItem& it = bb.item[blockIdx.y];
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= it.ns) return;
  const float* s = it.s;
  const GridMeta g = *it.meta;
  int64_t begin;
  const int b = cloud_of(it.lengths, bb.batch, i, begin);
  if (b >= bb.batch) {
    it.pt_cell[i] = -1;
    return;
  }
  int cx, cy, cz;
  cell_of(g, s[3 * i], s[3 * i + 1], s[3 * i + 2], cx, cy, cz);
  const int c = b * g.cells_per_cloud + (cz * g.dim[1] + cy) * g.dim[0] + cx;
  it.pt_cell[i] = c;
  it.pt_slot[i] = atomicAdd(&it.cell_count[c], 1);
}

// points per chunk of 4096 consecutive cells (block b = chunk b).  A plain reduction: accumulating these in the count kernel
// would be tens of thousands of atomics on a handful of addresses (neighbouring cells share a chunk).
__global__ __launch_bounds__(256) void rn_chunk_kernel(RnBuildBatch bb) {
  const RnBuildItem& it = bb.item[blockIdx.y];
  const int ncell = it.meta->cells_per_cloud * bb.batch;
  const int c0 = (static_cast<int>(blockIdx.x) << kChunkShift);
  if (c0 >= ncell) return;  // (uniform over the workgroup)
  __shared__ int wsum[4];
  int mine = 0;
  for (int k = threadIdx.x; k < (1 << kChunkShift); k += 256) mine += c0 + k < ncell ? it.cell_count[c0 + k] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) it.meta->chunk_sum[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// cell_start = exclusive prefix of cell_count in cell order: block b owns the 4096 cells of chunk b (16 per thread); its
// base is the sum of the chunks before it
__global__ __launch_bounds__(256) void rn_alloc_kernel(RnBuildBatch bb) {
  const RnBuildItem& it = bb.item[blockIdx.y];
  const int ncell = it.meta->cells_per_cloud * bb.batch;
  if ((static_cast<int>(blockIdx.x) << kChunkShift) >= ncell) return;  // (uniform over the workgroup)
  __shared__ int part[256];
  __shared__ int wsum[4];
  const int tid = threadIdx.x;
  const int c0 = (static_cast<int>(blockIdx.x) << kChunkShift) + 16 * tid;
  int cnt[16], mine = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    cnt[k] = c0 + k < ncell ? it.cell_count[c0 + k] : 0;
    mine += cnt[k];
  }
  part[tid] = mine;
  static_assert(kChunks == 256, "one thread per chunk sum below");
  int before = tid < static_cast<int>(blockIdx.x) ? it.meta->chunk_sum[tid] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if ((tid & 63) == 0) wsum[tid >> 6] = before;
  __syncthreads();
  const int base = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  for (int o = 1; o < 256; o <<= 1) {  // inclusive prefix of the per-thread sums
    const int add = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  int run = base + part[tid] - mine;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (c0 + k < ncell) it.cell_start[c0 + k] = run;
    run += cnt[k];
  }
}

// member lists: the indices of a cell's points, in arrival order
__global__ void rn_scatter_kernel(RnBuildBatch bb) {
  const RnBuildItem& it = bb.item[blockIdx.y];
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= it.ns) return;
  const int c = it.pt_cell[i];
  if (c < 0) return;
  it.cell_list[it.cell_start[c] + it.pt_slot[i]] = static_cast<int>(i);
}

// records in (cell, index) order: a point's place inside its cell is the number of members with a smaller index.
// Cells of up to 32 members (every cell of a voxel-subsampled level) are counted by the point's own lane; the members of a
// crowded cell -- raw scans, coincident points, the kMaxCells clamp on a huge extent -- are counted by the whole wavefront for one
// such point at a time (coalesced reads of the member list, 64 comparisons per trip): n / 64 trips per point instead of n
// dependent loads, so a degenerate cloud of 32 k points in ONE cell costs 16 M coalesced reads, not 10^9 serial ones.
__global__ void rn_rank_kernel(RnBuildBatch bb) {
  const RnBuildItem& it = bb.item[blockIdx.y];
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int c = i < it.ns ? it.pt_cell[i] : -1;
  const int start = c >= 0 ? it.cell_start[c] : 0, n = c >= 0 ? it.cell_count[c] : 0;
  int rank = 0;
  if (n <= 32)
    for (int k = 0; k < n; ++k) rank += it.cell_list[start + k] < static_cast<int>(i) ? 1 : 0;
  unsigned long long crowded = __builtin_amdgcn_ballot_w64(n > 32);
  while (crowded) {  // (wavefront-uniform)
    const int src = __builtin_ctzll(crowded);
    crowded &= crowded - 1;
    const int s0 = __builtin_amdgcn_readlane(start, src), sn = __builtin_amdgcn_readlane(n, src);
    const int si = __builtin_amdgcn_readlane(static_cast<int>(i), src);
    int cnt = 0;
    for (int k = lane; k < sn; k += 64) cnt += it.cell_list[s0 + k] < si ? 1 : 0;
    cnt = wave_sum_i(cnt);
    if (lane == src) rank = cnt;
  }
  if (c < 0) return;
  const float* s = it.s;
  float4 v;
  v.x = s[3 * i];
  v.y = s[3 * i + 1];
  v.z = s[3 * i + 2];
  v.w = __int_as_float(static_cast<int>(i));
  it.sorted[start + rank] = v;
}

// broadcast of lane `i` (wave-uniform) through scalar registers
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int i) {
  const unsigned lo = __builtin_amdgcn_readlane(static_cast<int>(v & 0xffffffffull), i);
  const unsigned hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), i);
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

#ifdef RDM_RN_TIMING
__device__ unsigned* rdm_rn_clk;  // tools/rn_phase_lab.hip: [queries][8] shader clocks per phase
#define RN_T0() unsigned long long rn_t = __builtin_amdgcn_s_memtime()
#define RN_PHASE(k) do { const unsigned long long now = __builtin_amdgcn_s_memtime(); if (lane == 0) rdm_rn_clk[qi * 8 + (k)] = static_cast<unsigned>(now - rn_t); rn_t = now; } while (0)
#else
#define RN_T0() do { } while (0)
#define RN_PHASE(k) do { } while (0)
#endif
// One query by one wavefront (all 64 lanes call it; no block-level barrier inside).  K: CAP keys, seg_start_w /
// seg_pref_w: 28 ints each, all private to the wavefront (LDS).
struct RnQueryArgs {
  const float* q;
  int64_t nq, ns;
  const int64_t* q_lengths;
  int batch;
  float radius;
  const GridMeta* meta;
  const int* cell_count;
  const int* cell_start;
  const float4* sorted;
  int width;
  int64_t* out_idx;
  int out32;  // rows of out_idx are int32 (the engine's internal tables) instead of int64
  int32_t* out_counts;
  int32_t* out_max;
  int32_t* status;
  unsigned char* redo;
};
// Bitonic sort of K[0, n) (n <= 1024, ascending) by one wavefront; K must hold the next power of two of n entries.
__device__ __forceinline__ void rn_sort_keys(unsigned long long* K, int n, int lane) {
  int p2 = 1;
  while (p2 < n) p2 <<= 1;
  for (int i = n + lane; i < p2; i += 64) K[i] = ~0ull;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // every stage reads all of its pairs, then writes them (plain LDS accesses batched by the compiler; wavefront-scope
  // fences order the stages)
  for (int k = 2; k <= p2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      unsigned long long lo[8], hi[8];
      const int pairs = p2 >> 1;  // pair t -> i = 2*j*(t / j) + (t % j), partner i + j (bit form below)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = lane + 64 * u;
        if (t < pairs) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // j is a power of two
          lo[u] = K[i];
          hi[u] = K[i + j];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = lane + 64 * u;
        if (t < pairs) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const bool up = (i & k) == 0;
          if ((lo[u] > hi[u]) == up) {
            K[i] = hi[u];
            K[i + j] = lo[u];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
}

// Walks the candidates of a query (the points of its 27 cells, `total` of them) 128 per step and calls
// f(hit, key) for both candidates of every lane -- uniformly, so that f may use wave ballots.
template <typename F>
__device__ __forceinline__ void rn_scan_candidates(const float4* sorted, const int* seg_start_w, const int* seg_pref_w, int total,
                                                   float qx, float qy, float qz, float r2, int lane, F&& f) {
  for (int base = 0; base < total; base += 128) {
    float4 p[2];
    int tt[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      tt[u] = base + 64 * u + lane;
      if (tt[u] < total) {
        int seg = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1)
          if (seg + step < 27 && seg_pref_w[seg + step] <= tt[u]) seg += step;
        p[u] = sorted[seg_start_w[seg] + (tt[u] - seg_pref_w[seg])];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bool hit = false;
      unsigned long long key = 0;
      if (tt[u] < total) {
        const float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
        float d2 = dx * dx;
        d2 = d2 + dy * dy;
        d2 = d2 + dz * dz;
        hit = d2 < r2;
        key = (static_cast<unsigned long long>(__float_as_uint(d2)) << 32) | static_cast<unsigned>(__float_as_int(p[u].w));
      }
      f(hit, key);
    }
  }
}

// A neighbourhood with more hits than the 1024-key buffer (the reference has no such limit:
// radius_neighbors_cpu.cpp:36-64 returns every neighbour).  The row is produced in rounds of at most 1024 columns: a
// radix select over the 64-bit (d2, index) keys -- 8-bit digits, one rescan of the candidates per digit, stopping as soon as
// the keys up to the current prefix fit the buffer -- finds the next batch of smallest keys, which is collected, sorted
// and written; the next round continues above the last key written.  Keys are unique (the index is part of the key).
// K: 1024 keys, hist: 256 ints, both private to the wavefront.
__device__ __forceinline__ void rn_dense_row(const RnQueryArgs& a, int64_t qi, int lane, unsigned long long* K, int* hist,
                                             const int* seg_start_w, const int* seg_pref_w, int total, int count, float qx,
                                             float qy, float qz, float r2) {
  const int out_n = count < a.width ? count : a.width;
  const long long row = qi * static_cast<long long>(a.width);  // element offset of the row (int64 or int32 elements: st_index)
  unsigned long long lower = 0;  // keys <= lower are already written (valid once emitted > 0)
  int emitted = 0;
  while (emitted < out_n) {
    const int want = out_n - emitted < 1024 ? out_n - emitted : 1024;
    const bool have_lower = emitted > 0;
    unsigned long long prefix = 0, upper = 0;
    int below = 0, n = 0;
    for (int shift = 56;; shift -= 8) {
      for (int i = lane; i < 256; i += 64) hist[i] = 0;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      rn_scan_candidates(a.sorted, seg_start_w, seg_pref_w, total, qx, qy, qz, r2, lane, [&](bool hit, unsigned long long key) {
        const bool in_prefix = shift == 56 || (key >> (shift + 8)) == (prefix >> (shift + 8));
        if (hit && in_prefix && (!have_lower || key > lower)) atomicAdd(&hist[static_cast<int>(key >> shift) & 255], 1);
      });
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // the digit whose bin holds the want-th remaining key: lane l owns bins 4l .. 4l+3
      int h[4], sum = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = hist[4 * lane + j];
        sum += h[j];
      }
      int inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      const int target = want - below;  // >= 1, and the bins hold at least that many keys
      const unsigned long long owner = __ballot(inc - sum < target && target <= inc);
      const int src = __builtin_ctzll(owner);
      int digit = 0, before = inc - sum, in_bin = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (in_bin == 0) {
          if (before + h[j] >= target) {
            digit = 4 * lane + j;
            in_bin = h[j];
          } else {
            before += h[j];
          }
        }
      }
      digit = __shfl(digit, src, 64);
      before = __shfl(before, src, 64);
      in_bin = __shfl(in_bin, src, 64);
      below += before;
      prefix |= static_cast<unsigned long long>(digit) << shift;
      if (below + in_bin <= 1024 || shift == 0) {
        upper = shift == 0 ? prefix : (prefix | ((1ull << shift) - 1ull));
        n = below + in_bin;
        break;
      }
    }
    // collect the keys in (lower, upper], sort, write
    int cnt = 0;
    rn_scan_candidates(a.sorted, seg_start_w, seg_pref_w, total, qx, qy, qz, r2, lane, [&](bool hit, unsigned long long key) {
      const bool take = hit && key <= upper && (!have_lower || key > lower);
      const unsigned long long m = __ballot(take);
      if (take) {
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < 1024) K[pos] = key;
      }
      cnt += __popcll(m);
    });
    n = cnt < 1024 ? cnt : 1024;  // (= below + in_bin)
    rn_sort_keys(K, n, lane);
    const int m = n < out_n - emitted ? n : out_n - emitted;
    for (int c = lane; c < m; c += 64) st_index(a.out_idx, row + emitted + c, static_cast<long long>(K[c] & 0xffffffffull), a.out32);
    lower = K[m - 1];
    emitted += m;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // K is rewritten by the next round
    __builtin_amdgcn_wave_barrier();
  }
  for (int c = out_n + lane; c < a.width; c += 64) st_index(a.out_idx, row + c, a.ns, a.out32);
}

template <int CAP>
__device__ __forceinline__ void rn_query_one(const RnQueryArgs& a, int64_t qi, int lane, unsigned long long* K, int* seg_start_w,
                                             int* seg_pref_w, int only_redo, int* hist = nullptr) {
  const float* q = a.q;
  const int64_t ns = a.ns;
  const int64_t* q_lengths = a.q_lengths;
  const int batch = a.batch;
  const float radius = a.radius;
  const GridMeta* meta = a.meta;
  const int* cell_count = a.cell_count;
  const int* cell_start = a.cell_start;
  const float4* sorted = a.sorted;
  const int width = a.width;
  int64_t* out_idx = a.out_idx;
  int32_t* out_counts = a.out_counts;
  int32_t* out_max = a.out_max;
  int32_t* status = a.status;
  unsigned char* redo = a.redo;
  RN_T0();
  const GridMeta g = *meta;
  if (radius * g.inv_cell > 1.0f) {  // the grid was built for a smaller radius: 27 cells would miss neighbours
    if (lane == 0) atomicExch(status, 2);
    return;
  }
  const float r2 = radius * radius;
  const float qx = q[3 * qi], qy = q[3 * qi + 1], qz = q[3 * qi + 2];
  int64_t begin;
  const int b = cloud_of(q_lengths, batch, qi, begin);

  int count = 0, total = 0;
  if (b < batch) {
    int cx, cy, cz;
    cell_of(g, qx, qy, qz, cx, cy, cz);
    RN_PHASE(0);
    // lanes 0..26 own one neighbouring cell each
    int my_n = 0, my_start = 0;
    if (lane < 27) {
      const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
      const int x = cx + dx, y = cy + dy, z = cz + dz;
      if (x >= 0 && x < g.dim[0] && y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2]) {
        const int c = b * g.cells_per_cloud + (z * g.dim[1] + y) * g.dim[0] + x;
        my_n = cell_count[c];
        my_start = cell_start[c];
      }
    }
    int inc = my_n;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane < 27) {
      seg_start_w[lane] = my_start;
      seg_pref_w[lane] = inc - my_n;
    }
    total = __shfl(inc, 26, 64);
    if (lane == 27) seg_pref_w[27] = total;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

    RN_PHASE(1);
    // two candidates per lane and step: both record loads are in flight before either is tested
    for (int base = 0; base < total; base += 128) {
      bool hit[2] = {false, false};
      unsigned long long key[2] = {0, 0};
      float4 p[2];
      int tt[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        tt[u] = base + 64 * u + lane;
        if (tt[u] < total) {
          int seg = 0;
#pragma unroll
          for (int step = 16; step > 0; step >>= 1)
            if (seg + step < 27 && seg_pref_w[seg + step] <= tt[u]) seg += step;
          p[u] = sorted[seg_start_w[seg] + (tt[u] - seg_pref_w[seg])];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (tt[u] < total) {
          const float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
          float d2 = dx * dx;
          d2 = d2 + dy * dy;
          d2 = d2 + dz * dz;
          hit[u] = d2 < r2;
          key[u] = (static_cast<unsigned long long>(__float_as_uint(d2)) << 32) |
                   static_cast<unsigned>(__float_as_int(p[u].w));
        }
        const unsigned long long m = __ballot(hit[u]);
        if (hit[u]) {
          const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
          if (pos < CAP) K[pos] = key[u];
        }
        count += __popcll(m);
      }
    }
  }
  RN_PHASE(2);
  if (lane == 0) {
    if (out_counts) out_counts[qi] = count;
    // one address for all queries: same-address atomics cost ~12 ns each, so only raise it when needed
    if (out_max && count > ld_agent(out_max)) atomicMax(out_max, count);
  }
  if (count > CAP) {  // the sorted prefix cannot be produced from a truncated buffer
    if (CAP >= 1024 && hist != nullptr) {  // large-buffer pass: neighbourhoods beyond the buffer are produced in rounds
      if (width > 0) rn_dense_row(a, qi, lane, K, hist, seg_start_w, seg_pref_w, total, count, qx, qy, qz, r2);
      return;
    }
    if (lane == 0) {
      if (only_redo || !redo) atomicExch(status, 1);
      else redo[qi] = 1;
    }
    if (width > 0 && !only_redo && redo) return;  // the large-buffer pass writes this row
  } else if (lane == 0 && redo && !only_redo) {
    redo[qi] = 0;
  }
  if (width <= 0) return;

  const int n = count < CAP ? count : CAP;
  if (n <= 128) {
    // rank sort in registers (the common case: a neighbourhood holds 40-100 points): each lane keeps up to two
    // keys and counts the keys smaller than its own while every key is broadcast once with v_readlane -- no LDS
    // traffic, no barriers; keys are unique (the index is part of the key), so ranks are a permutation
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long k0 = lane < n ? K[lane] : ~0ull;
    const unsigned long long k1 = 64 + lane < n ? K[64 + lane] : ~0ull;
    if (width == 1) {
      // one column (the nearest-neighbour tables of a plain engine run: up-sampling, functional.py:6-22): the smallest key is
      // the first of the sorted row -- six exchange steps instead of the n-step rank sort
      unsigned long long mn = k0 < k1 ? k0 : k1;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = (static_cast<unsigned long long>(static_cast<unsigned>(__shfl_xor(static_cast<int>(mn >> 32), o, 64))) << 32) |
                                         static_cast<unsigned>(__shfl_xor(static_cast<int>(mn & 0xffffffffull), o, 64));
        mn = other < mn ? other : mn;
      }
      if (lane == 0) st_index(out_idx, qi, n > 0 ? static_cast<long long>(mn & 0xffffffffull) : ns, a.out32);
      return;
    }
    int r0 = 0, r1 = 0;
    if (n <= 64) {  // one key per lane: half the comparisons
      for (int i = 0; i < n; ++i) r0 += readlane64(k0, i) < k0 ? 1 : 0;
    } else {
      for (int i = 0; i < 64; ++i) {
        const unsigned long long ki = readlane64(k0, i);
        r0 += ki < k0 ? 1 : 0;
        r1 += ki < k1 ? 1 : 0;
      }
      for (int i = 64; i < n; ++i) {
        const unsigned long long ki = readlane64(k1, i - 64);
        r0 += ki < k0 ? 1 : 0;
        r1 += ki < k1 ? 1 : 0;
      }
    }
    RN_PHASE(3);
    // every lane knows the final column of its keys: write the row directly (pads behind the n-th column)
    const long long row = qi * static_cast<long long>(width);
    if (lane < n && r0 < width) st_index(out_idx, row + r0, static_cast<long long>(k0 & 0xffffffffull), a.out32);
    if (64 + lane < n && r1 < width) st_index(out_idx, row + r1, static_cast<long long>(k1 & 0xffffffffull), a.out32);
    for (int c = n + lane; c < width; c += 64) st_index(out_idx, row + c, ns, a.out32);
    RN_PHASE(4);
    return;
  }
  rn_sort_keys(K, n, lane);
  // offset of this cloud's supports is already folded in (indices are global rows)
  const long long row = qi * static_cast<long long>(width);
  for (int c = lane; c < width; c += 64) st_index(out_idx, row + c, c < n ? static_cast<long long>(K[c] & 0xffffffffull) : ns, a.out32);
}

// First pass (CAP = 256, one wavefront per query, every query) and the stand-alone second pass (CAP = 1024, only the
// queries whose `redo` flag the first pass set).
template <int CAP>
__global__ __launch_bounds__(64 * kWavesPerBlock) void rn_query_kernel(RnQueryArgs a, int only_redo) {
  __shared__ unsigned long long keys[kWavesPerBlock][CAP];
  __shared__ int seg_start[kWavesPerBlock][28];
  __shared__ int seg_pref[kWavesPerBlock][28];
  __shared__ int hist[CAP >= 1024 ? kWavesPerBlock : 1][CAP >= 1024 ? 256 : 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t qi = blockIdx.x * static_cast<int64_t>(kWavesPerBlock) + wave;
  if (qi >= a.nq) return;  // whole wave exits together
  if (only_redo && !a.redo[qi]) return;  // second pass: only the queries that overflowed the small buffer
  rn_query_one<CAP>(a, qi, lane, keys[wave], seg_start[wave], seg_pref[wave], only_redo, CAP >= 1024 ? hist[wave] : nullptr);
}

// The second pass of SEVERAL searches in one launch (the engine defers it: 14 searches per scan pair, and a launch
// that finds nothing to do still costs its dispatch).  blockIdx.y = search; the wavefronts of a search walk its redo
// flags 64 at a time and take the flagged queries one by one.
constexpr int kRedoMax = 16;
struct RnRedoBatch {
  RnQueryArgs item[kRedoMax];
  int first_block[kRedoMax + 1];  // first pass in one launch: workgroup range of every search
  int n;
};
// First pass of SEVERAL searches in one launch: a 1-D grid, workgroup -> (search, query block) through first_block.
__global__ __launch_bounds__(64 * kWavesPerBlock) void rn_query_multi_kernel(RnRedoBatch b) {
  __shared__ unsigned long long keys[kWavesPerBlock][256];
  __shared__ int seg_start[kWavesPerBlock][28];
  __shared__ int seg_pref[kWavesPerBlock][28];
  int it = 0;
#pragma unroll
  for (int k = 1; k < kRedoMax; ++k) it += (k < b.n && static_cast<int>(blockIdx.x) >= b.first_block[k]) ? 1 : 0;
  const RnQueryArgs& a = b.item[it];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t qi = (blockIdx.x - b.first_block[it]) * static_cast<int64_t>(kWavesPerBlock) + wave;
  if (qi >= a.nq) return;
  rn_query_one<256>(a, qi, lane, keys[wave], seg_start[wave], seg_pref[wave], 0);
}
__global__ __launch_bounds__(64 * kWavesPerBlock) void rn_redo_multi_kernel(RnRedoBatch b) {
  __shared__ unsigned long long keys[kWavesPerBlock][1024];
  __shared__ int seg_start[kWavesPerBlock][28];
  __shared__ int seg_pref[kWavesPerBlock][28];
  __shared__ int hist[kWavesPerBlock][256];
  const RnQueryArgs& a = b.item[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t base = (blockIdx.x * static_cast<int64_t>(kWavesPerBlock) + wave) * 64; base < a.nq; base += nwaves * 64) {
    unsigned long long todo = __ballot(base + lane < a.nq && a.redo[base + lane] != 0);
    while (todo) {
      const int k = __builtin_ctzll(todo);
      todo &= todo - 1;
      rn_query_one<1024>(a, base + k, lane, keys[wave], seg_start[wave], seg_pref[wave], 1, hist[wave]);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the LDS buffers are reused by the next query
      __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace

namespace {
struct GridViews {
  GridMeta* meta;
  int* cell_count;
  int* cell_start;
  int* pt_cell;
  int* pt_slot;
  int* cell_list;
  float4* sorted;
};
bool carve_grid(rdm::Arena& ar, int64_t n_s, GridViews* g) {
  const size_t ns = static_cast<size_t>(n_s > 0 ? n_s : 1);
  g->meta = ar.take<GridMeta>(1);
  g->cell_count = ar.take<int>(kMaxCells);
  g->cell_start = ar.take<int>(kMaxCells);
  g->pt_cell = ar.take<int>(ns);
  g->pt_slot = ar.take<int>(ns);
  g->cell_list = ar.take<int>(ns);
  g->sorted = ar.take<float4>(ns);
  return ar.ok;
}
}  // namespace

extern "C" size_t rdm_radius_grid_workspace_bytes(int64_t n_s) {
  rdm::Arena a(nullptr, 0);
  GridViews g;
  carve_grid(a, n_s, &g);
  return a.off;
}

int rdm::radius_grid_build_multi(int n, const float* const* s_points, const int64_t* n_s, const int64_t* const* s_lengths,
                                 int batch, const float* radius, void* const* grid_ws, const size_t* grid_ws_bytes, void* stream) {
  RDM_REQUIRE(n >= 1 && n <= kBuildMax && batch > 0 && batch <= kMaxBatch, "rdm_radius_grid_build: bad arguments (n=%d batch=%d)", n, batch);
  RnBuildBatch bb;
  bb.n = n; bb.batch = batch;
  int64_t max_ns = 0;
  for (int k = 0; k < n; ++k) {
    RDM_REQUIRE(s_lengths[k] && grid_ws[k], "rdm_radius_grid_build: null pointer");
    RDM_REQUIRE(n_s[k] >= 0 && n_s[k] < (1ll << 31) && radius[k] > 0.f, "rdm_radius_grid_build: bad arguments (n_s=%lld)", (long long)n_s[k]);
    RDM_REQUIRE(n_s[k] == 0 || s_points[k], "rdm_radius_grid_build: null points");
    Arena ar(grid_ws[k], grid_ws_bytes[k]);
    GridViews g;
    if (!carve_grid(ar, n_s[k], &g)) {
      set_error("rdm_radius_grid_build: workspace too small (%zu < %zu bytes)", grid_ws_bytes[k], ar.off);
      return RDM_ERR_WORKSPACE;
    }
    bb.item[k] = RnBuildItem{s_points[k], n_s[k], s_lengths[k], radius[k], g.meta, g.cell_count, g.cell_start, g.pt_cell, g.pt_slot, g.cell_list, g.sorted};
    max_ns = std::max(max_ns, n_s[k]);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(rn_bbox_kernel, dim3(n), dim3(1024), 0, st, bb);
  hipLaunchKernelGGL(rn_zero_kernel, dim3(256, n), dim3(256), 0, st, bb);
  if (max_ns > 0) {
    const int blocks = static_cast<int>(ceil_div<int64_t>(max_ns, 256));
    hipLaunchKernelGGL(rn_count_kernel, dim3(blocks, n), dim3(256), 0, st, bb);
    hipLaunchKernelGGL(rn_chunk_kernel, dim3(kChunks, n), dim3(256), 0, st, bb);
    hipLaunchKernelGGL(rn_alloc_kernel, dim3(kChunks, n), dim3(256), 0, st, bb);
    hipLaunchKernelGGL(rn_scatter_kernel, dim3(blocks, n), dim3(256), 0, st, bb);
    hipLaunchKernelGGL(rn_rank_kernel, dim3(blocks, n), dim3(256), 0, st, bb);
  }
  return launch_status("radius grid build");
}

namespace {
// "Grid" of ONE cell per cloud: the queries then test every point of their cloud (brute force), which for a few hundred
// points -- the NMS search over the shifted superpoints, vote.py:24-31 -- is cheaper than the seven launches of a real grid.
__global__ void rn_trivial_grid_kernel(const float* s, int64_t ns, const int64_t* lengths, int batch, GridMeta* meta, int* cell_count,
                                       int* cell_start, float4* sorted) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i == 0) {
    for (int d = 0; d < 3; ++d) {
      meta->org[d] = 0.f;
      meta->dim[d] = 1;
    }
    meta->inv_cell = 0.f;  // cell_of -> (0, 0, 0) for every point; any radius passes the "grid built for this radius" test
    meta->cells_per_cloud = 1;
    meta->total = 0;
    int64_t acc = 0;
    for (int b = 0; b < batch; ++b) {
      cell_start[b] = static_cast<int>(acc);
      cell_count[b] = static_cast<int>(lengths[b]);
      acc += lengths[b];
    }
  }
  if (i < ns) sorted[i] = make_float4(s[3 * i], s[3 * i + 1], s[3 * i + 2], __int_as_float(static_cast<int>(i)));
}
}  // namespace

int rdm::radius_grid_build_trivial(const float* s_points, int64_t n_s, const int64_t* s_lengths, int batch, void* grid_ws,
                                   size_t grid_ws_bytes, void* stream) {
  RDM_REQUIRE(s_points && s_lengths && grid_ws && n_s > 0 && n_s < (1ll << 31) && batch > 0 && batch <= kMaxBatch,
              "radius_grid_build_trivial: bad arguments");
  Arena ar(grid_ws, grid_ws_bytes);
  GridViews g;
  if (!carve_grid(ar, n_s, &g)) {
    set_error("radius_grid_build_trivial: workspace too small (%zu < %zu bytes)", grid_ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(rn_trivial_grid_kernel, dim3(static_cast<unsigned>(ceil_div<int64_t>(n_s, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), s_points, n_s, s_lengths, batch, g.meta, g.cell_count, g.cell_start, g.sorted);
  return launch_status("rn_trivial_grid_kernel");
}

extern "C" int rdm_radius_grid_build(const float* s_points, int64_t n_s, const int64_t* s_lengths, int batch,
                                     float radius, void* grid_ws, size_t grid_ws_bytes, void* stream) {
  RDM_REQUIRE(s_lengths && grid_ws, "rdm_radius_grid_build: null pointer");
  return rdm::radius_grid_build_multi(1, &s_points, &n_s, &s_lengths, batch, &radius, &grid_ws, &grid_ws_bytes, stream);
}

namespace {
// shared by the C entry point and the engine's deferred form: first pass now; the second pass now (queue == null) or
// appended to `queue` for rdm::radius_redo_flush
int grid_query(void* grid_ws, size_t grid_ws_bytes, int64_t n_s, const float* q_points, int64_t n_q, const int64_t* q_lengths,
               int batch, float radius, int width, int64_t* out_idx, int32_t* out_counts, int32_t* out_max, int32_t* status,
               unsigned char* redo, RnRedoBatch* queue, int out32, hipStream_t st) {
  using namespace rdm;
  RDM_REQUIRE(grid_ws && q_lengths && status, "rdm_radius_grid_query: null pointer");
  RDM_REQUIRE(n_q >= 0 && batch > 0 && batch <= kMaxBatch && radius > 0.f, "rdm_radius_grid_query: bad arguments");
  RDM_REQUIRE(width >= 0 && (width == 0 || out_idx), "rdm_radius_grid_query: width/out_idx mismatch");
  if (n_q == 0) return RDM_OK;
  RDM_REQUIRE(q_points, "rdm_radius_grid_query: null points");
  Arena gar(grid_ws, grid_ws_bytes);
  GridViews g;
  RDM_REQUIRE(carve_grid(gar, n_s, &g), "rdm_radius_grid_query: grid workspace size does not match n_s");
  RnQueryArgs a;
  a.q = q_points; a.nq = n_q; a.ns = n_s; a.q_lengths = q_lengths; a.batch = batch; a.radius = radius; a.meta = g.meta;
  a.cell_count = g.cell_count; a.cell_start = g.cell_start; a.sorted = g.sorted; a.width = width; a.out_idx = out_idx;
  a.out32 = out32 ? 1 : 0;
  a.out_counts = out_counts; a.out_max = out_max; a.status = status; a.redo = redo;
  const int qblocks = static_cast<int>(ceil_div<int64_t>(n_q, kWavesPerBlock));
  // small per-wavefront buffers keep many wavefronts resident; the rare query with more than 256
  // neighbours is redone by the large-buffer instance
  if (queue && queue->n < kRedoMax && width > 0) {  // both passes run at the flush, together with the other searches
    queue->first_block[queue->n + 1] = queue->first_block[queue->n] + qblocks;
    queue->item[queue->n++] = a;
    return RDM_OK;
  }
  hipLaunchKernelGGL(rn_query_kernel<256>, dim3(qblocks), dim3(64 * kWavesPerBlock), 0, st, a, 0);
  if (width > 0) hipLaunchKernelGGL(rn_query_kernel<1024>, dim3(qblocks), dim3(64 * kWavesPerBlock), 0, st, a, 1);
  return launch_status("rn_query_kernel");
}
}  // namespace

size_t rdm::radius_redo_queue_bytes() { return sizeof(RnRedoBatch); }
void rdm::radius_redo_queue_reset(void* queue) {
  static_cast<RnRedoBatch*>(queue)->n = 0;
  static_cast<RnRedoBatch*>(queue)->first_block[0] = 0;
}
int rdm::radius_grid_query_deferred(void* grid_ws, size_t grid_ws_bytes, int64_t n_s, const float* q_points, int64_t n_q,
                                    const int64_t* q_lengths, int batch, float radius, int width, int64_t* out_idx,
                                    int32_t* out_counts, int32_t* out_max, int32_t* status, unsigned char* redo_flags,
                                    void* queue, int i32, void* stream) {
  RDM_REQUIRE(redo_flags && queue, "radius_grid_query_deferred: null redo storage");
  return grid_query(grid_ws, grid_ws_bytes, n_s, q_points, n_q, q_lengths, batch, radius, width, out_idx, out_counts, out_max,
                    status, redo_flags, static_cast<RnRedoBatch*>(queue), i32, static_cast<hipStream_t>(stream));
}
int rdm::radius_redo_flush(void* queue, void* stream) {
  RnRedoBatch* b = static_cast<RnRedoBatch*>(queue);
  if (b->n == 0) return RDM_OK;
  RDM_DUP_LOOP("rn") {  // (first pass + its large-buffer second pass: the pair is idempotent)
    hipLaunchKernelGGL(rn_query_multi_kernel, dim3(b->first_block[b->n]), dim3(64 * kWavesPerBlock), 0, static_cast<hipStream_t>(stream), *b);
    // 64 workgroups (256 wavefronts) per search: two flag sweeps of 64 queries per wavefront cover a 32 k-point level
    hipLaunchKernelGGL(rn_redo_multi_kernel, dim3(64, b->n), dim3(64 * kWavesPerBlock), 0, static_cast<hipStream_t>(stream), *b);
  }
  b->n = 0;
  b->first_block[0] = 0;
  return launch_status("rn_query_multi_kernel");
}

extern "C" int rdm_radius_grid_query(void* grid_ws, size_t grid_ws_bytes, int64_t n_s, const float* q_points,
                                     int64_t n_q, const int64_t* q_lengths, int batch, float radius, int width,
                                     int64_t* out_idx, int32_t* out_counts, int32_t* out_max, int32_t* status,
                                     void* ws, size_t ws_bytes, void* stream) {
  using namespace rdm;
  Arena ar(ws, ws_bytes);
  unsigned char* redo = ar.take<unsigned char>(static_cast<size_t>(n_q > 0 ? n_q : 1));
  if (!ar.ok) {
    set_error("rdm_radius_grid_query: workspace too small (%zu < %zu bytes)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  return grid_query(grid_ws, grid_ws_bytes, n_s, q_points, n_q, q_lengths, batch, radius, width, out_idx, out_counts, out_max,
                    status, redo, nullptr, 0, static_cast<hipStream_t>(stream));
}

extern "C" const float* rdm_radius_grid_records(void* grid_ws, size_t grid_ws_bytes, int64_t n_s) {
  rdm::Arena ar(grid_ws, grid_ws_bytes);
  GridViews g;
  if (!carve_grid(ar, n_s, &g)) return nullptr;
  return reinterpret_cast<const float*>(g.sorted);
}

namespace {
// Histogram of per-query neighbour counts (calibration of the neighbour limits): np.bincount(counts,
// minlength = hist_n)[:hist_n] accumulated into hist.  Per-block LDS histogram, then one global atomic per
// non-empty bin, so the many queries with similar counts do not serialise on HBM atomics.
constexpr int kHistLds = 1024;
__global__ __launch_bounds__(256) void rn_histogram_kernel(const int32_t* __restrict__ counts, int64_t n,
                                                            int32_t* __restrict__ hist, int hist_n) {
  __shared__ int32_t local[kHistLds];
  for (int i = threadIdx.x; i < kHistLds; i += 256) local[i] = 0;
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
    const int32_t c = counts[i];
    if (c >= 0 && c < hist_n) atomicAdd(&local[c], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < hist_n; i += 256)
    if (local[i]) atomicAdd(hist + i, local[i]);
}

}  // namespace

extern "C" int rdm_neighbor_histogram(const int32_t* counts, int64_t n, int32_t* hist, int hist_n, void* stream) {
  RDM_REQUIRE(n >= 0 && hist_n > 0 && hist_n <= kHistLds, "rdm_neighbor_histogram: hist_n must be in 1..%d", kHistLds);
  if (n == 0) return 0;
  RDM_REQUIRE(counts && hist, "rdm_neighbor_histogram: null pointer");
  const int64_t blocks = std::min<int64_t>(ceil_div<int64_t>(n, 256 * 8), 1024);
  hipLaunchKernelGGL(rn_histogram_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), counts, n, hist, hist_n);
  return launch_status("rn_histogram_kernel");
}

extern "C" size_t rdm_radius_neighbors_workspace_bytes(int64_t n_q, int64_t n_s, int batch) {
  (void)batch;
  return rdm_radius_grid_workspace_bytes(n_s) + rdm::align_up(static_cast<size_t>(n_q > 0 ? n_q : 1));
}

extern "C" int rdm_radius_neighbors(const float* q_points, int64_t n_q, const float* s_points,
                                    int64_t n_s, const int64_t* q_lengths, const int64_t* s_lengths,
                                    int batch, float radius, int width, int64_t* out_idx,
                                    int32_t* out_counts, int32_t* out_max, int32_t* status, void* ws,
                                    size_t ws_bytes, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(q_lengths && s_lengths && status, "rdm_radius_neighbors: null pointer");
  RDM_REQUIRE(width >= 0 && (width == 0 || out_idx), "rdm_radius_neighbors: width/out_idx mismatch");
  RDM_REQUIRE(radius > 0.f, "rdm_radius_neighbors: radius must be positive");
  if (n_q == 0) return RDM_OK;
  const size_t gbytes = rdm_radius_grid_workspace_bytes(n_s);
  if (ws == nullptr || ws_bytes < gbytes) {
    set_error("rdm_radius_neighbors: workspace too small (%zu < %zu bytes)", ws_bytes, gbytes);
    return RDM_ERR_WORKSPACE;
  }
  if (int e = rdm_radius_grid_build(s_points, n_s, s_lengths, batch, radius, ws, gbytes, stream)) return e;
  return rdm_radius_grid_query(ws, gbytes, n_s, q_points, n_q, q_lengths, batch, radius, width, out_idx, out_counts, out_max,
                               status, static_cast<char*>(ws) + gbytes, ws_bytes - gbytes, stream);
}
