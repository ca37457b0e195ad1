// a10 NMS, a11 point-to-node grouping, a12 coarse superpoint matching.
//
// All three produce INDEX tensors, so their arithmetic restates the reference's CPU arithmetic
// literally (no contraction except where the reference's BLAS fuses):
//   pairwise_distance (modules/ops/pairwise_distance.py:4-31): d = (|x|^2 - 2*xy) + |y|^2, clamp 1e-12,
//   with |x|^2 = (x0*x0 + x1*x1) + x2*x2 and xy = fma(x2,y2, fma(x1,y1, x0*y0)) -- the order the
//   reference's sgemm uses for k = 3 (verified bit-for-bit against torch.matmul when the goldens
//   were generated, tests/golden/oracle_vs_reference.json).
#pragma clang fp contract(off)

#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "internal.h"
#include "lockstep.h"

namespace {

using namespace rdm;

// ---------------------------------------------------------------------------------------------
// NMS (rdmnet/vote/vote.py:13-40): greedy in index order, keep[i] = !any(keep[nbr(i)]) evaluated
// when only indices < i have been decided.  keep[i] therefore depends on lower-index neighbours
// only, which makes it the lexicographically-first maximal independent set; it is resolved in
// parallel rounds (a node is final once a lower neighbour is kept, or all lower neighbours are
// final and none is kept).  One workgroup; rounds <= longest dependency chain.
__device__ __forceinline__ void nms_kernel_body(const dim3 blockIdx, const dim3 gridDim, const int64_t* idx, int n, int h, int ldi,
                                                   const int32_t* width, unsigned char* keep) {
  (void)blockIdx; (void)gridDim;
  extern __shared__ unsigned char state[];  // 0 undecided, 1 kept, 2 suppressed
  __shared__ int pending;
  int H = h;
  if (width) H = min(H, *width);
  for (int i = threadIdx.x; i < n; i += blockDim.x) state[i] = 0;
  __syncthreads();
  for (int round = 0; round <= n; ++round) {
    if (threadIdx.x == 0) pending = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      if (state[i] != 0) continue;
      bool any_kept = false, any_open = false;
      for (int c = 0; c < H; ++c) {
        const int64_t j = idx[static_cast<int64_t>(i) * ldi + c];
        if (j < 0 || j >= i) continue;  // later nodes are still False when i is visited (vote.py:36-38)
        const unsigned char s = state[j];
        any_kept |= (s == 1);
        any_open |= (s == 0);
      }
      // a racing read of a neighbour that flips this round only delays the decision by one round
      if (any_kept) state[i] = 2;
      else if (!any_open) state[i] = 1;
      else pending = 1;
    }
    __syncthreads();
    if (pending == 0) break;
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) keep[i] = state[i] == 1 ? 1 : 0;
}
__global__ __launch_bounds__(1024) void nms_kernel(const int64_t* idx, int n, int h, int ldi,
                                                   const int32_t* width, unsigned char* keep) { nms_kernel_body(blockIdx, gridDim, idx, n, h, ldi, width, keep); }


// order-preserving compaction of kept rows: dst row = number of kept rows before it
__device__ __forceinline__ void compact_index_kernel_body(const dim3 blockIdx, const dim3 gridDim, const unsigned char* keep, int begin, int end,
                                                             int32_t* order, int32_t* count) {
  (void)blockIdx; (void)gridDim;
  __shared__ int wsum[17];
  __shared__ int base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int i0 = begin; i0 < end; i0 += blockDim.x) {
    const int i = i0 + threadIdx.x;
    const int flag = (i < end && keep[i]) ? 1 : 0;
    int inc = flag;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = 0;
      for (int k = 0; k < 16; ++k) {
        const int t = wsum[k];
        wsum[k] = acc;
        acc += t;
      }
      wsum[16] = acc;
    }
    __syncthreads();
    if (flag) order[base + wsum[w] + inc - 1] = i;
    __syncthreads();
    if (threadIdx.x == 0) base += wsum[16];
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = base;
}
__global__ __launch_bounds__(1024) void compact_index_kernel(const unsigned char* keep, int begin, int end,
                                                             int32_t* order, int32_t* count) { compact_index_kernel_body(blockIdx, gridDim, keep, begin, end, order, count); }


// Both clouds' compactions in one launch (workgroup b: rows [b ? n_ref : 0, b ? n : n_ref), indices from order + begin,
// count in counts[b]), and -- for the engine's size read-back -- a copy of the `mirror_words` status words at mirror_src
// into mirror_dst (mapped host memory) with the two counts stored directly into their slots: no separate copy launch.
__device__ __forceinline__ void compact_index_pair_kernel_body(const dim3 blockIdx, const dim3 gridDim, const unsigned char* keep, int n_ref, int n, int32_t* order,
                                                                  int32_t* counts, const int32_t* mirror_src,
                                                                  int32_t* mirror_dst, int mirror_words) {
  (void)blockIdx; (void)gridDim;
  __shared__ int wsum[17];
  __shared__ int base;
  const int begin = blockIdx.x ? n_ref : 0, end = blockIdx.x ? n : n_ref;
  int32_t* out = order + begin;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int i0 = begin; i0 < end; i0 += blockDim.x) {
    const int i = i0 + threadIdx.x;
    const int flag = (i < end && keep[i]) ? 1 : 0;
    int inc = flag;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = 0;
      for (int k = 0; k < 16; ++k) {
        const int t = wsum[k];
        wsum[k] = acc;
        acc += t;
      }
      wsum[16] = acc;
    }
    __syncthreads();
    if (flag) out[base + wsum[w] + inc - 1] = i;
    __syncthreads();
    if (threadIdx.x == 0) base += wsum[16];
    __syncthreads();
  }
  const int slot = static_cast<int>(counts - mirror_src) + blockIdx.x;  // where this count lives among the status words
  if (threadIdx.x == 0) {
    counts[blockIdx.x] = base;
    if (mirror_dst) mirror_dst[slot] = base;
  }
  if (mirror_dst && blockIdx.x == 0) {
    const int s0 = static_cast<int>(counts - mirror_src);
    for (int t = threadIdx.x; t < mirror_words; t += blockDim.x)
      if (t != s0 && t != s0 + 1) mirror_dst[t] = mirror_src[t];
  }
}
__global__ __launch_bounds__(1024) void compact_index_pair_kernel(const unsigned char* keep, int n_ref, int n, int32_t* order,
                                                                  int32_t* counts, const int32_t* mirror_src,
                                                                  int32_t* mirror_dst, int mirror_words) { compact_index_pair_kernel_body(blockIdx, gridDim, keep, n_ref, n, order, counts, mirror_src, mirror_dst, mirror_words); }


// ---------------------------------------------------------------------------------------------
// point_to_node_partition (modules/ops/pointcloud_partition.py:60-107)
__device__ __forceinline__ float ref_sq_dist(float x0, float x1, float x2, float xn, float y0, float y1,
                                             float y2, float yn) {
  const float xy = fmaf(x2, y2, fmaf(x1, y1, x0 * y0));
  float d = (xn - 2.f * xy) + yn;
  return d < 1e-12f ? 1e-12f : d;
}

// one thread per point: owner = argmin over nodes (first minimum), d_own = that distance
__device__ __forceinline__ void p2n_assign_body(unsigned block_x, const float* points, int n, const float* nodes, int m, int32_t* owner,
                                                float* d_own, int32_t* node_count) {
  extern __shared__ float sn[];  // [m][4]: x, y, z, |node|^2
  for (int j = threadIdx.x; j < m; j += blockDim.x) {
    const float a = nodes[3 * j], b = nodes[3 * j + 1], c = nodes[3 * j + 2];
    sn[4 * j] = a;
    sn[4 * j + 1] = b;
    sn[4 * j + 2] = c;
    sn[4 * j + 3] = (a * a + b * b) + c * c;
  }
  __syncthreads();
  const int i = block_x * blockDim.x + threadIdx.x;  // (the caller's block index: a grouped launch carries its own, lockstep.h)
  if (i >= n) return;
  const float y0 = points[3 * i], y1 = points[3 * i + 1], y2 = points[3 * i + 2];
  const float yn = (y0 * y0 + y1 * y1) + y2 * y2;
  float best = INFINITY;
  int arg = 0;
  for (int j = 0; j < m; ++j) {
    const float d = ref_sq_dist(sn[4 * j], sn[4 * j + 1], sn[4 * j + 2], sn[4 * j + 3], y0, y1, y2, yn);
    if (d < best) {
      best = d;
      arg = j;
    }
  }
  owner[i] = arg;
  d_own[i] = best;
  atomicAdd(&node_count[arg], 1);
}

__device__ __forceinline__ void p2n_assign_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* points, int n, const float* nodes, int m,
                                                         int32_t* owner, float* d_own, int32_t* node_count) {
  (void)gridDim;
  p2n_assign_body(blockIdx.x, points, n, nodes, m, owner, d_own, node_count);
}
__global__ __launch_bounds__(256) void p2n_assign_kernel(const float* points, int n, const float* nodes, int m,
                                                         int32_t* owner, float* d_own, int32_t* node_count) { p2n_assign_kernel_body(blockIdx, gridDim, points, n, nodes, m, owner, d_own, node_count); }


// Both clouds of a pair in one launch (blockIdx.y = cloud): the two assignments are independent.
struct P2nSeg {
  const float* points;
  int n;
  const float* nodes;
  int m;
  int32_t* owner;
  float* d_own;
  int32_t* node_count;
  int64_t* knn_idx;
  unsigned char* knn_mask;
  unsigned char* node_mask;
};
struct P2nPair {
  P2nSeg seg[2];
  int k;
  int32_t* status;
};
__device__ __forceinline__ void p2n_assign_pair_kernel_body(const dim3 blockIdx, const dim3 gridDim, P2nPair p) {
  (void)blockIdx; (void)gridDim;
  const P2nSeg& s = p.seg[blockIdx.y];
  if (static_cast<int>(blockIdx.x) * 256 >= s.n) return;  // whole workgroup
  p2n_assign_body(blockIdx.x, s.points, s.n, s.nodes, s.m, s.owner, s.d_own, s.node_count);
}
__global__ __launch_bounds__(256) void p2n_assign_pair_kernel(P2nPair p) { p2n_assign_pair_kernel_body(blockIdx, gridDim, p); }


// one wavefront per node: its points sorted by (d, index), first k kept (topk largest=False)
template <int CAP>
__device__ __forceinline__ void p2n_select_body(unsigned long long* keys, int node, const int32_t* owner, const float* d_own, int n,
                                                int m, int k, const int32_t* node_count, int64_t* knn_idx,
                                                unsigned char* knn_mask, unsigned char* node_mask, int32_t* status) {
  const int lane = threadIdx.x;
  const int cnt = node_count[node];
  if (lane == 0) node_mask[node] = cnt > 0 ? 1 : 0;
  if (cnt > CAP && lane == 0) atomicExch(status, 1);
  volatile unsigned long long* K = keys;
  int filled = 0;
  for (int base = 0; base < n; base += 256) {  // four owner loads in flight per step (the scan is latency-bound)
    int own[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 64 * u + lane;
      own[u] = i < n ? owner[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 64 * u + lane;
      const bool mine = own[u] == node;
      const unsigned long long mm = __ballot(mine);
      if (mine) {
        const int pos = filled + __popcll(mm & ((1ull << lane) - 1ull));
        if (pos < CAP) K[pos] = (static_cast<unsigned long long>(__float_as_uint(d_own[i])) << 32) | static_cast<unsigned>(i);
      }
      filled += __popcll(mm);
    }
  }
  const int cn = filled < CAP ? filled : CAP;
  int p2 = 1;
  while (p2 < cn) p2 <<= 1;
  for (int i = cn + lane; i < p2; i += 64) K[i] = ~0ull;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int kk = 2; kk <= p2; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = lane; i < p2; i += 64) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = K[i], c = K[ixj];
          if ((a > c) == ((i & kk) == 0)) {
            K[i] = c;
            K[ixj] = a;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  for (int c = lane; c < k; c += 64) {
    const bool ok = c < cn;
    knn_idx[static_cast<int64_t>(node) * k + c] = ok ? static_cast<int64_t>(K[c] & 0xffffffffull) : n;
    knn_mask[static_cast<int64_t>(node) * k + c] = ok ? 1 : 0;
  }
}

template <int CAP>
__device__ __forceinline__ void p2n_select_kernel_body(const dim3 blockIdx, const dim3 gridDim, const int32_t* owner, const float* d_own, int n, int m,
                                                        int k, const int32_t* node_count, int64_t* knn_idx,
                                                        unsigned char* knn_mask, unsigned char* node_mask,
                                                        int32_t* status) {
  (void)blockIdx; (void)gridDim;
  __shared__ unsigned long long keys[CAP];
  p2n_select_body<CAP>(keys, blockIdx.x, owner, d_own, n, m, k, node_count, knn_idx, knn_mask, node_mask, status);
}
template <int CAP>
__global__ __launch_bounds__(64) void p2n_select_kernel(const int32_t* owner, const float* d_own, int n, int m,
                                                        int k, const int32_t* node_count, int64_t* knn_idx,
                                                        unsigned char* knn_mask, unsigned char* node_mask,
                                                        int32_t* status) { p2n_select_kernel_body<CAP>(blockIdx, gridDim, owner, d_own, n, m, k, node_count, knn_idx, knn_mask, node_mask, status); }

template <int CAP>
__device__ __forceinline__ void p2n_select_pair_kernel_body(const dim3 blockIdx, const dim3 gridDim, P2nPair p) {
  (void)blockIdx; (void)gridDim;
  __shared__ unsigned long long keys[CAP];
  const P2nSeg& s = p.seg[blockIdx.y];
  if (static_cast<int>(blockIdx.x) >= s.m) return;
  p2n_select_body<CAP>(keys, blockIdx.x, s.owner, s.d_own, s.n, s.m, p.k, s.node_count, s.knn_idx, s.knn_mask, s.node_mask, p.status);
}
template <int CAP>
__global__ __launch_bounds__(64) void p2n_select_pair_kernel(P2nPair p) { p2n_select_pair_kernel_body<CAP>(blockIdx, gridDim, p); }


// ---------------------------------------------------------------------------------------------
// SuperPointMatching (modules/geotransformer/superpoint_matching.py:14-61)
// scores = exp(-clamp(2 - 2*xy, 1e-12)) over non-empty nodes, rows/cols of empty nodes are 0
__device__ __forceinline__ void coarse_scores_kernel_body(const dim3 blockIdx, const dim3 gridDim, float* s, int m, int n, int ld, const unsigned char* rmask,
                                     const unsigned char* cmask) {
  (void)blockIdx; (void)gridDim;
  const int64_t t = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (t >= static_cast<int64_t>(m) * n) return;
  const int i = static_cast<int>(t / n), j = static_cast<int>(t % n);
  float v = 0.f;
  if (rmask[i] && cmask[j]) {
    float d = 2.0f - 2.0f * s[static_cast<int64_t>(i) * ld + j];
    d = d < 1e-12f ? 1e-12f : d;
    v = expf(-d);
  }
  s[static_cast<int64_t>(i) * ld + j] = v;
}
__global__ void coarse_scores_kernel(float* s, int m, int n, int ld, const unsigned char* rmask,
                                     const unsigned char* cmask) { coarse_scores_kernel_body(blockIdx, gridDim, s, m, n, ld, rmask, cmask); }

// rsum[i] = sum_j s[i,j] (one wavefront per row); csum[j] = sum_i s[i,j] (one thread per column)
__device__ __forceinline__ void coarse_rowsum_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* s, int m, int n, int ld, float* rsum) {
  (void)blockIdx; (void)gridDim;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= m) return;
  float acc = 0.f;
  for (int j = threadIdx.x & 63; j < n; j += 64) acc += s[static_cast<int64_t>(i) * ld + j];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) rsum[i] = acc;
}
__global__ void coarse_rowsum_kernel(const float* s, int m, int n, int ld, float* rsum) { coarse_rowsum_kernel_body(blockIdx, gridDim, s, m, n, ld, rsum); }

__device__ __forceinline__ void coarse_colsum_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* s, int m, int n, int ld, float* csum) {
  (void)blockIdx; (void)gridDim;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float acc = 0.f;  // rows are added in ascending order (as the reference's sum over dim 0), 8 loads in flight
  int i = 0;
  for (; i + 8 <= m; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = s[static_cast<int64_t>(i + u) * ld + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; i < m; ++i) acc += s[static_cast<int64_t>(i) * ld + j];
  csum[j] = acc;
}
__global__ void coarse_colsum_kernel(const float* s, int m, int n, int ld, float* csum) { coarse_colsum_kernel_body(blockIdx, gridDim, s, m, n, ld, csum); }

__device__ __forceinline__ void coarse_dual_kernel_body(const dim3 blockIdx, const dim3 gridDim, float* s, int m, int n, int ld, const float* rsum, const float* csum,
                                   const unsigned char* rmask, const unsigned char* cmask) {
  (void)blockIdx; (void)gridDim;
  const int64_t t = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (t >= static_cast<int64_t>(m) * n) return;
  const int i = static_cast<int>(t / n), j = static_cast<int>(t % n);
  float v = -1.f;  // empty nodes never enter the ranking (they are removed before topk in the reference)
  if (rmask[i] && cmask[j]) {
    const float x = s[static_cast<int64_t>(i) * ld + j];
    v = rsum ? (x / rsum[i]) * (x / csum[j]) : x;
  }
  s[static_cast<int64_t>(i) * ld + j] = v;
}
__global__ void coarse_dual_kernel(float* s, int m, int n, int ld, const float* rsum, const float* csum,
                                   const unsigned char* rmask, const unsigned char* cmask) { coarse_dual_kernel_body(blockIdx, gridDim, s, m, n, ld, rsum, csum, rmask, cmask); }


// ---- the same stage from the FEATURES, evaluated in fp64 (rdm_coarse_matching_features).
// Why fp64: the reference's top-k order is decided by relative score gaps down to 4e-7 (tests/golden/
// coarse_order_analysis.json); fp32 sums of ~300 terms in another order than the reference's BLAS / reduction move a
// score by that much, the exact value does not (the reference's own fp32 evaluation agrees with fp64 at every position
// on both golden cases).  The stage is tiny (m*n*d = 26 MFLOP), so exactness costs nothing measurable.
// One workgroup per 16 x 16 tile of pairs; feature rows staged in LDS (row stride d+1: conflict-free column walks).
__device__ __forceinline__ void coarse_scores64_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* fr, int ldr, int m, const float* fs, int lds_, int n,
                                                              int d, const unsigned char* rmask, const unsigned char* cmask,
                                                              double* s, int ld) {
  (void)blockIdx; (void)gridDim;
  extern __shared__ float tile[];
  float* a = tile;                    // [16][d+1]
  float* b = tile + 16 * (d + 1);     // [16][d+1]
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  for (int t = threadIdx.x; t < 16 * d; t += 256) {
    const int r = t / d, c = t % d;
    a[r * (d + 1) + c] = i0 + r < m ? fr[static_cast<int64_t>(i0 + r) * ldr + c] : 0.f;
    b[r * (d + 1) + c] = j0 + r < n ? fs[static_cast<int64_t>(j0 + r) * lds_ + c] : 0.f;
  }
  __syncthreads();
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15, i = i0 + ti, j = j0 + tj;
  if (i >= m || j >= n) return;
  double v = 0.0;
  if (rmask[i] && cmask[j]) {
    const float* pa = a + ti * (d + 1);
    const float* pb = b + tj * (d + 1);
    double acc0 = 0.0, acc1 = 0.0;
    int c = 0;
    for (; c + 2 <= d; c += 2) {
      acc0 = fma(static_cast<double>(pa[c]), static_cast<double>(pb[c]), acc0);
      acc1 = fma(static_cast<double>(pa[c + 1]), static_cast<double>(pb[c + 1]), acc1);
    }
    if (c < d) acc0 = fma(static_cast<double>(pa[c]), static_cast<double>(pb[c]), acc0);
    double dist = 2.0 - 2.0 * (acc0 + acc1);  // pairwise_distance(normalized=True), clamp(min=1e-12)
    dist = dist < 1e-12 ? 1e-12 : dist;
    v = exp(-dist);
  }
  s[static_cast<int64_t>(i) * ld + j] = v;
}
__global__ __launch_bounds__(256) void coarse_scores64_kernel(const float* fr, int ldr, int m, const float* fs, int lds_, int n,
                                                              int d, const unsigned char* rmask, const unsigned char* cmask,
                                                              double* s, int ld) { coarse_scores64_kernel_body(blockIdx, gridDim, fr, ldr, m, fs, lds_, n, d, rmask, cmask, s, ld); }

__device__ __forceinline__ void coarse_rowsum64_kernel_body(const dim3 blockIdx, const dim3 gridDim, const double* s, int m, int n, int ld, double* rsum) {
  (void)blockIdx; (void)gridDim;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= m) return;
  double acc = 0.0;
  for (int j = threadIdx.x & 63; j < n; j += 64) acc += s[static_cast<int64_t>(i) * ld + j];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) rsum[i] = acc;
}
__global__ void coarse_rowsum64_kernel(const double* s, int m, int n, int ld, double* rsum) { coarse_rowsum64_kernel_body(blockIdx, gridDim, s, m, n, ld, rsum); }

__device__ __forceinline__ void coarse_colsum64_kernel_body(const dim3 blockIdx, const dim3 gridDim, const double* s, int m, int n, int ld, double* csum) {
  (void)blockIdx; (void)gridDim;
  // one wavefront per 64 columns x a slice of rows would need a second pass; the matrix is small: 4 row slices per column
  // block reduced through LDS in a fixed order
  __shared__ double part[4][64];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
  double acc = 0.0;
  if (j < n)
    for (int i = slice; i < m; i += 4) acc += s[static_cast<int64_t>(i) * ld + j];
  part[slice][threadIdx.x & 63] = acc;
  __syncthreads();
  if (slice == 0 && j < n) csum[j] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
__global__ void coarse_colsum64_kernel(const double* s, int m, int n, int ld, double* csum) { coarse_colsum64_kernel_body(blockIdx, gridDim, s, m, n, ld, csum); }

__device__ __forceinline__ void coarse_dual64_kernel_body(const dim3 blockIdx, const dim3 gridDim, const double* s, int m, int n, int ld, const double* rsum, const double* csum,
                                     const unsigned char* rmask, const unsigned char* cmask, float* out, int ldo) {
  (void)blockIdx; (void)gridDim;
  const int64_t t = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (t >= static_cast<int64_t>(m) * n) return;
  const int i = static_cast<int>(t / n), j = static_cast<int>(t % n);
  float v = -1.f;  // empty nodes never enter the ranking
  if (rmask[i] && cmask[j]) {
    const double x = s[static_cast<int64_t>(i) * ld + j];
    v = static_cast<float>(rsum ? (x / rsum[i]) * (x / csum[j]) : x);
  }
  out[static_cast<int64_t>(i) * ldo + j] = v;
}
__global__ void coarse_dual64_kernel(const double* s, int m, int n, int ld, const double* rsum, const double* csum,
                                     const unsigned char* rmask, const unsigned char* cmask, float* out, int ldo) { coarse_dual64_kernel_body(blockIdx, gridDim, s, m, n, ld, rsum, csum, rmask, cmask, out, ldo); }


// Global top-k (k <= 1024) of an m x n matrix, descending, ties by ascending flat index.
// One workgroup: three radix-select passes on the float bit pattern, then a bitonic sort.
// `gate` (optional, a TopkState): run only if its fallback flag is set, i.e. as the fallback of the multi-workgroup path below.
__device__ __forceinline__ void topk_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* s, int m, int n, int ld, int k,
                                                    int64_t* out_row, int64_t* out_col, float* out_val,
                                                    int32_t* out_count, const unsigned* gate) {
  (void)blockIdx; (void)gridDim;
  if (gate && gate[5] == 0) return;  // TopkState::fallback
  __shared__ unsigned hist[4096];
  __shared__ unsigned long long cand[2048];
  __shared__ unsigned sh_prefix, sh_need, sh_ncand;
  __shared__ unsigned wcnt[17];
  const int total = m * n;  // (host side guarantees m * n < 2^31)
  const int sw_w = threadIdx.x >> 6, sw_lane = threadIdx.x & 63;
// one sweep over the matrix without integer divisions: wavefront w takes rows w, w+16, ..., lanes stride the columns
#define RDM_SWEEP(BODY)                                                           \
  for (int row = sw_w; row < m; row += 16) {                                      \
    const float* pr = s + static_cast<int64_t>(row) * ld;                         \
    const int tb = row * n;                                                       \
    for (int c0 = 0; c0 < n; c0 += 512) { /* 8 independent loads in flight */     \
      float vv[8];                                                                \
      _Pragma("unroll") for (int q = 0; q < 8; ++q) {                             \
        const int col = c0 + 64 * q + sw_lane;                                    \
        vv[q] = col < n ? pr[col] : -1.f;                                         \
      }                                                                           \
      _Pragma("unroll") for (int q = 0; q < 8; ++q) {                             \
        const int col = c0 + 64 * q + sw_lane;                                    \
        if (col < n) {                                                            \
          const float v = vv[q];                                                  \
          const int t = tb + col;                                                 \
          (void)t;                                                                \
          BODY                                                                    \
        }                                                                         \
      }                                                                           \
    }                                                                             \
  }
  // count eligible (>= 0) entries
  if (threadIdx.x == 0) sh_ncand = 0;
  __syncthreads();
  unsigned elig = 0;
  RDM_SWEEP(elig += v >= 0.f ? 1 : 0;)
  atomicAdd(&sh_ncand, elig);
  __syncthreads();
  const int kk = min<int>(k, static_cast<int>(sh_ncand));
  __syncthreads();
  if (kk == 0) {
    if (threadIdx.x == 0) *out_count = 0;
    return;
  }
  // radix select the kk-th largest bit pattern (non-negative floats order like unsigned ints)
  unsigned prefix = 0, need = kk;  // among values whose high bits == prefix, find the need-th largest
  const int shifts[3] = {20, 8, 0};
  const int bits[3] = {12, 12, 8};
  unsigned mask_hi = 0;
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << bits[pass];
    for (int i = threadIdx.x; i < nb; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    RDM_SWEEP(if (v >= 0.f) {
      const unsigned u = __float_as_uint(v);
      if ((u & mask_hi) == prefix) atomicAdd(&hist[(u >> shifts[pass]) & (nb - 1)], 1u);
    })
    __syncthreads();
    // find the bin holding the need-th largest value: wavefront 0 scans from the top, 64 bins per lane
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x, per = nb / 64;  // nb is 4096 or 256
      const int hi = nb - 1 - lane * per;           // this lane owns bins hi, hi-1, ..., hi-per+1
      unsigned mine = 0;
      for (int q = 0; q < per; ++q) mine += hist[hi - q];
      unsigned incl = mine;                          // inclusive prefix over lanes (higher bins first)
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      const unsigned before = incl - mine;
      if (before < need && incl >= need) {           // exactly one lane
        unsigned acc = before;
        int b = hi;
        for (; b > hi - per + 1; --b) {
          if (acc + hist[b] >= need) break;
          acc += hist[b];
        }
        sh_prefix = prefix | (static_cast<unsigned>(b) << shifts[pass]);
        sh_need = need - acc;
      }
    }
    __syncthreads();
    prefix = sh_prefix;
    need = sh_need;
    mask_hi |= static_cast<unsigned>(nb - 1) << shifts[pass];
    __syncthreads();
  }
  // prefix = exact bit pattern of the kk-th largest value; `need` of the entries equal to it are kept
  // (lowest flat indices first); everything greater is kept.
  if (threadIdx.x == 0) sh_ncand = 0;
  __syncthreads();
  // greater-than entries: any order (sorted below)
  RDM_SWEEP(if (v >= 0.f) {
    const unsigned u = __float_as_uint(v);
    if (u > prefix) {
      const unsigned pos = atomicAdd(&sh_ncand, 1u);
      if (pos < 2048) cand[pos] = (static_cast<unsigned long long>(~u) << 32) | static_cast<unsigned>(t);
    }
  })
  __syncthreads();
  const unsigned n_gt = sh_ncand;
  __syncthreads();
  // equal entries: the `need` lowest flat indices among them.  Usually the value is unique (need == 1, one
  // entry): one more sweep lists the equal entries in `hist` (reused), a short selection keeps the lowest
  // indices; only a value with more than 4096 duplicates falls back to the ordered chunk scan.
  if (threadIdx.x == 0) sh_need = 0;  // reused: number of equal entries listed / taken so far
  __syncthreads();
  RDM_SWEEP(if (v >= 0.f && __float_as_uint(v) == prefix) {
    const unsigned pos = atomicAdd(&sh_need, 1u);
    if (pos < 4096) hist[pos] = static_cast<unsigned>(t);
  })
  __syncthreads();
  const unsigned n_eq = sh_need;
  __syncthreads();
  if (n_eq <= 4096) {
    // rank of every listed index among the list (indices are distinct); ranks < need are kept
    for (unsigned i = threadIdx.x; i < n_eq; i += blockDim.x) {
      const unsigned mine = hist[i];
      unsigned r = 0;
      for (unsigned j = 0; j < n_eq; ++j) r += hist[j] < mine ? 1u : 0u;
      if (r < need) cand[n_gt + r] = (static_cast<unsigned long long>(~prefix) << 32) | mine;
    }
    __syncthreads();
  } else {
  if (threadIdx.x == 0) sh_need = 0;
  __syncthreads();
  for (int t0 = 0; t0 < total && sh_need < need; t0 += blockDim.x) {
    const int t = t0 + threadIdx.x;
    bool eq = false;
    if (t < total) {
      const unsigned ur = static_cast<unsigned>(t) / static_cast<unsigned>(n);
      const float v = s[static_cast<int64_t>(ur) * ld + (static_cast<unsigned>(t) - ur * static_cast<unsigned>(n))];
      eq = v >= 0.f && __float_as_uint(v) == prefix;
    }
    // ordered rank inside the chunk
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long bm = __ballot(eq);
    if (lane == 0) wcnt[w] = __popcll(bm);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned acc = 0;
      for (int q = 0; q < 16; ++q) {
        const unsigned c = wcnt[q];
        wcnt[q] = acc;
        acc += c;
      }
      wcnt[16] = acc;
    }
    __syncthreads();
    if (eq) {
      const unsigned r = sh_need + wcnt[w] + __popcll(bm & ((1ull << lane) - 1ull));
      if (r < need) cand[n_gt + r] = (static_cast<unsigned long long>(~prefix) << 32) | static_cast<unsigned>(t);
    }
    __syncthreads();
    if (threadIdx.x == 0) sh_need += wcnt[16];
    __syncthreads();
  }
  }
#undef RDM_SWEEP
  // sort kk candidates: key = (~value bits, flat index) ascending == value desc, index asc
  int p2 = 1;
  while (p2 < kk) p2 <<= 1;
  for (int i = kk + threadIdx.x; i < p2; i += blockDim.x) cand[i] = ~0ull;
  __syncthreads();
  for (int a = 2; a <= p2; a <<= 1)
    for (int j = a >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < p2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = cand[i], y = cand[ixj];
          if ((x > y) == ((i & a) == 0)) {
            cand[i] = y;
            cand[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < kk; i += blockDim.x) {
    const unsigned long long c = cand[i];
    const unsigned flat = static_cast<unsigned>(c & 0xffffffffull);
    out_row[i] = flat / n;
    out_col[i] = flat % n;
    out_val[i] = __uint_as_float(~static_cast<unsigned>(c >> 32));
  }
  if (threadIdx.x == 0) *out_count = kk;
}
__global__ __launch_bounds__(1024) void topk_kernel(const float* s, int m, int n, int ld, int k,
                                                    int64_t* out_row, int64_t* out_col, float* out_val,
                                                    int32_t* out_count, const unsigned* gate) { topk_kernel_body(blockIdx, gridDim, s, m, n, ld, k, out_row, out_col, out_val, out_count, gate); }


}  // namespace

extern "C" int rdm_nms(const int64_t* idx, int64_t n, int64_t h, int64_t ldi, const int32_t* width,
                       uint8_t* keep, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(idx && keep && n >= 0 && h > 0, "rdm_nms: bad arguments");
  RDM_REQUIRE(n <= 150000, "rdm_nms: at most 150000 nodes (LDS-resident state)");
  if (n == 0) return RDM_OK;
  ::rdm::launch<nms_kernel_body, nms_kernel, 1024>(dim3(1), static_cast<size_t>(n), static_cast<hipStream_t>(stream),
                     idx, static_cast<int>(n), static_cast<int>(h), static_cast<int>(ldi), width, keep);
  return launch_status("nms_kernel");
}

extern "C" int rdm_compact_indices(const uint8_t* keep, int64_t begin, int64_t end, int32_t* order,
                                   int32_t* count, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(keep && order && count && begin >= 0 && end >= begin, "rdm_compact_indices: bad arguments");
  ::rdm::launch<compact_index_kernel_body, compact_index_kernel, 1024>(dim3(1), 0, static_cast<hipStream_t>(stream), keep,
                     static_cast<int>(begin), static_cast<int>(end), order, count);
  return launch_status("compact_index_kernel");
}

int rdm::compact_indices_pair(const uint8_t* keep, int64_t n_ref, int64_t n, int32_t* order, int32_t* counts,
                              const int32_t* mirror_src, int32_t* mirror_dst, int mirror_words, void* stream) {
  RDM_REQUIRE(keep && order && counts && n_ref >= 0 && n >= n_ref, "compact_indices_pair: bad arguments");
  RDM_REQUIRE(!mirror_dst || (mirror_src && counts >= mirror_src && counts + 2 <= mirror_src + mirror_words),
              "compact_indices_pair: the counts must lie inside the mirrored words");
  ::rdm::launch<compact_index_pair_kernel_body, compact_index_pair_kernel, 1024>(dim3(2), 0, static_cast<hipStream_t>(stream), keep,
                     static_cast<int>(n_ref), static_cast<int>(n), order, counts, mirror_src, mirror_dst, mirror_words);
  return launch_status("compact_index_pair_kernel");
}

extern "C" size_t rdm_point_to_node_workspace_bytes(int64_t n_points, int64_t n_nodes) {
  rdm::Arena a(nullptr, 0);
  a.take<int32_t>(n_points > 0 ? n_points : 1);
  a.take<float>(n_points > 0 ? n_points : 1);
  a.take<int32_t>(n_nodes > 0 ? n_nodes : 1);
  return a.off;
}

extern "C" int rdm_point_to_node(const float* points, int64_t n_points, const float* nodes, int64_t n_nodes,
                                 int k, int64_t* knn_idx, uint8_t* knn_mask, uint8_t* node_mask,
                                 int32_t* status, void* ws, size_t ws_bytes, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(points && nodes && knn_idx && knn_mask && node_mask && status, "rdm_point_to_node: null pointer");
  RDM_REQUIRE(n_points > 0 && n_nodes > 0 && k > 0 && n_nodes <= 8192,
              "rdm_point_to_node: bad sizes (points=%lld nodes=%lld)", (long long)n_points, (long long)n_nodes);
  Arena ar(ws, ws_bytes);
  int32_t* owner = ar.take<int32_t>(n_points);
  float* d_own = ar.take<float>(n_points);
  int32_t* node_count = ar.take<int32_t>(n_nodes);
  if (!ar.ok) {
    set_error("rdm_point_to_node: workspace too small (%zu < %zu)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  fill_words<int32_t>(node_count, n_nodes, 0, st);
  ::rdm::launch<p2n_assign_kernel_body, p2n_assign_kernel, 256>(dim3(ceil_div<int64_t>(n_points, 256)), static_cast<size_t>(n_nodes) * 16, st, points, static_cast<int>(n_points), nodes,
                     static_cast<int>(n_nodes), owner, d_own, node_count);
  ::rdm::launch<p2n_select_kernel_body<4096>, p2n_select_kernel<4096>, 64>(dim3(static_cast<unsigned>(n_nodes)), 0, st, owner, d_own,
                     static_cast<int>(n_points), static_cast<int>(n_nodes), k, node_count, knn_idx, knn_mask,
                     node_mask, status);
  return launch_status("point_to_node kernels");
}

// rdm_point_to_node for the two clouds of a pair with one set of launches (same results as two calls).
extern "C" int rdm_point_to_node_pair(const float* points_a, int64_t n_a, const float* nodes_a, int64_t m_a,
                                      const float* points_b, int64_t n_b, const float* nodes_b, int64_t m_b, int k,
                                      int64_t* knn_idx_a, uint8_t* knn_mask_a, uint8_t* node_mask_a, int64_t* knn_idx_b,
                                      uint8_t* knn_mask_b, uint8_t* node_mask_b, int32_t* status, void* ws, size_t ws_bytes,
                                      void* stream) {
  using namespace rdm;
  RDM_REQUIRE(points_a && nodes_a && points_b && nodes_b && knn_idx_a && knn_mask_a && node_mask_a && knn_idx_b && knn_mask_b &&
                  node_mask_b && status, "rdm_point_to_node_pair: null pointer");
  RDM_REQUIRE(n_a > 0 && m_a > 0 && n_b > 0 && m_b > 0 && k > 0 && m_a <= 8192 && m_b <= 8192,
              "rdm_point_to_node_pair: bad sizes (points=%lld/%lld nodes=%lld/%lld)", (long long)n_a, (long long)n_b,
              (long long)m_a, (long long)m_b);
  Arena ar(ws, ws_bytes);
  P2nPair p;
  p.k = k; p.status = status;
  int32_t* counts = ar.take<int32_t>(m_a + m_b);  // contiguous: one fill
  p.seg[0] = P2nSeg{points_a, static_cast<int>(n_a), nodes_a, static_cast<int>(m_a), ar.take<int32_t>(n_a), ar.take<float>(n_a), counts,
                    knn_idx_a, knn_mask_a, node_mask_a};
  p.seg[1] = P2nSeg{points_b, static_cast<int>(n_b), nodes_b, static_cast<int>(m_b), ar.take<int32_t>(n_b), ar.take<float>(n_b),
                    counts ? counts + m_a : nullptr, knn_idx_b, knn_mask_b, node_mask_b};
  if (!ar.ok) {
    set_error("rdm_point_to_node_pair: workspace too small (%zu < %zu)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n_max = n_a > n_b ? n_a : n_b, m_max = m_a > m_b ? m_a : m_b;
  fill_words<int32_t>(counts, m_a + m_b, 0, st);
  ::rdm::launch<p2n_assign_pair_kernel_body, p2n_assign_pair_kernel, 256>(dim3(ceil_div<int64_t>(n_max, 256), 2), static_cast<size_t>(m_max) * 16, st, p);
  ::rdm::launch<p2n_select_pair_kernel_body<4096>, p2n_select_pair_kernel<4096>, 64>(dim3(static_cast<unsigned>(m_max), 2), 0, st, p);
  return launch_status("point_to_node kernels");
}

namespace {

// ---- multi-workgroup top-k: the single-workgroup kernel above needs ~130 us on one CU for a 330 x 320 matrix
// (five sweeps over 105 k scores); here the sweeps run on the whole GPU:
//   1. histogram of the top 12 bits of every eligible score (LDS per block, flushed with integer atomics);
//      one wavefront finds the bin that holds the k-th largest score
//   2. the same for the next 12 bits of the scores inside that bin (real score matrices put thousands there)
//   3. every score at or above that 24-bit prefix is appended to a candidate list (any order)
//   4. one workgroup sorts the (at most kTopkCand) candidates by (score desc, flat index asc) and emits the first k
// The result is the same function of the matrix as topk_kernel (ties by ascending flat index); more than
// kTopkCand candidates (thousands of scores sharing their top 12 bits) fall back to topk_kernel.
constexpr int kTopkBins = 4096, kTopkCand = 2048, kTopkRows = 8;
struct TopkState {  // device scratch
  unsigned eligible;   // scores >= 0
  unsigned bin0;       // threshold bin of bits [31:20]
  unsigned need1;      // how many of bin0's scores are still needed
  unsigned bin1;       // threshold bin of bits [19:8] inside bin0
  unsigned n_cand;     // candidates appended
  unsigned fallback;   // 1: candidate list overflowed, topk_kernel must run
  unsigned pad[2];
};

// LEVEL 0: histogram of bits [31:20] of all eligible scores; LEVEL 1: bits [19:8] of the scores inside bin0
template <int LEVEL>
__device__ __forceinline__ void topk_hist_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* __restrict__ s, int m, int n, int ld,
                                                         const TopkState* __restrict__ st, unsigned* __restrict__ ghist) {
  (void)blockIdx; (void)gridDim;
  __shared__ unsigned hist[kTopkBins];
  for (int i = threadIdx.x; i < kTopkBins; i += 256) hist[i] = 0;
  __syncthreads();
  const unsigned bin0 = LEVEL == 1 ? st->bin0 : 0;
  const int r0 = blockIdx.x * kTopkRows, r1 = min(m, r0 + kTopkRows);
  for (int row = r0 + (threadIdx.x >> 6); row < r1; row += 4)
    for (int col = threadIdx.x & 63; col < n; col += 64) {
      const float v = s[static_cast<int64_t>(row) * ld + col];
      if (v < 0.f) continue;
      const unsigned u = __float_as_uint(v);
      if (LEVEL == 0) atomicAdd(&hist[u >> 20], 1u);
      else if ((u >> 20) == bin0) atomicAdd(&hist[(u >> 8) & 0xfffu], 1u);
    }
  __syncthreads();
  for (int i = threadIdx.x; i < kTopkBins; i += 256)
    if (hist[i]) atomicAdd(&ghist[i], hist[i]);
}
template <int LEVEL>
__global__ __launch_bounds__(256) void topk_hist_kernel(const float* __restrict__ s, int m, int n, int ld,
                                                         const TopkState* __restrict__ st, unsigned* __restrict__ ghist) { topk_hist_kernel_body<LEVEL>(blockIdx, gridDim, s, m, n, ld, st, ghist); }


// the bin that holds the need-th largest entry of the histogram, and how many entries of that bin are needed
template <int LEVEL>
__device__ __forceinline__ void topk_pick_kernel_body(const dim3 blockIdx, const dim3 gridDim, const unsigned* __restrict__ ghist, int k, TopkState* st) {
  (void)blockIdx; (void)gridDim;
  const int lane = threadIdx.x, per = kTopkBins / 64;
  const int hi = kTopkBins - 1 - lane * per;  // this lane owns bins hi, hi-1, ..., hi-per+1
  unsigned mine = 0;
  for (int q = 0; q < per; ++q) mine += ghist[hi - q];
  unsigned incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  const unsigned total = __shfl(incl, 63, 64);
  const unsigned need = LEVEL == 0 ? min(static_cast<unsigned>(k), total) : st->need1;
  if (LEVEL == 0 && lane == 0) {
    st->eligible = total;
    st->n_cand = 0;
    st->fallback = 0;
    if (need == 0) {  // nothing eligible: no bin qualifies
      st->bin0 = kTopkBins;
      st->need1 = 0;
    }
  }
  if (LEVEL == 1 && lane == 0 && need == 0) st->bin1 = kTopkBins;
  const unsigned before = incl - mine;
  if (need > 0 && before < need && incl >= need) {  // exactly one lane
    unsigned acc = before;
    int b = hi;
    for (; b > hi - per + 1; --b) {
      if (acc + ghist[b] >= need) break;
      acc += ghist[b];
    }
    if (LEVEL == 0) {
      st->bin0 = static_cast<unsigned>(b);
      st->need1 = need - acc;
    } else {
      st->bin1 = static_cast<unsigned>(b);
    }
  }
}
template <int LEVEL>
__global__ __launch_bounds__(64) void topk_pick_kernel(const unsigned* __restrict__ ghist, int k, TopkState* st) { topk_pick_kernel_body<LEVEL>(blockIdx, gridDim, ghist, k, st); }


// every score above the 24-bit threshold prefix (bin0, bin1) or sharing it is a candidate
__device__ __forceinline__ void topk_collect_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* __restrict__ s, int m, int n, int ld, TopkState* st,
                                                            unsigned long long* __restrict__ cand) {
  (void)blockIdx; (void)gridDim;
  const unsigned thr = (st->bin0 << 12) | (st->bin1 & 0xfffu);
  const bool none = st->bin0 >= kTopkBins;
  const int r0 = blockIdx.x * kTopkRows, r1 = min(m, r0 + kTopkRows);
  for (int row = r0 + (threadIdx.x >> 6); row < r1; row += 4)
    for (int col = threadIdx.x & 63; col < n; col += 64) {
      const float v = s[static_cast<int64_t>(row) * ld + col];
      if (v < 0.f || none) continue;
      const unsigned u = __float_as_uint(v);
      if ((u >> 8) >= thr) {
        const unsigned pos = atomicAdd(&st->n_cand, 1u);
        if (pos < kTopkCand) cand[pos] = (static_cast<unsigned long long>(~u) << 32) | static_cast<unsigned>(row * n + col);
      }
    }
}
__global__ __launch_bounds__(256) void topk_collect_kernel(const float* __restrict__ s, int m, int n, int ld, TopkState* st,
                                                            unsigned long long* __restrict__ cand) { topk_collect_kernel_body(blockIdx, gridDim, s, m, n, ld, st, cand); }


__device__ __forceinline__ void topk_final_kernel_body(const dim3 blockIdx, const dim3 gridDim, const unsigned long long* __restrict__ cand, TopkState* st, int n,
                                                           int k, int64_t* out_row, int64_t* out_col, float* out_val,
                                                           int32_t* out_count) {
  (void)blockIdx; (void)gridDim;
  __shared__ unsigned long long keys[kTopkCand];
  const unsigned nc = st->n_cand;
  if (nc > kTopkCand) {  // thousands of scores share their top 24 bits: the single-workgroup kernel runs instead
    if (threadIdx.x == 0) st->fallback = 1;
    return;
  }
  const int kk = static_cast<int>(min(static_cast<unsigned>(k), st->eligible));
  int p2 = 1;
  while (p2 < static_cast<int>(nc)) p2 <<= 1;
  for (int i = threadIdx.x; i < p2; i += 1024) keys[i] = i < static_cast<int>(nc) ? cand[i] : ~0ull;
  __syncthreads();
  for (int a = 2; a <= p2; a <<= 1)
    for (int j = a >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < p2; i += 1024) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = keys[i], y = keys[ixj];
          if ((x > y) == ((i & a) == 0)) {
            keys[i] = y;
            keys[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < kk; i += 1024) {
    const unsigned long long c = keys[i];
    const unsigned flat = static_cast<unsigned>(c & 0xffffffffull);
    out_row[i] = flat / n;
    out_col[i] = flat % n;
    out_val[i] = __uint_as_float(~static_cast<unsigned>(c >> 32));
  }
  if (threadIdx.x == 0) *out_count = kk;
}
__global__ __launch_bounds__(1024) void topk_final_kernel(const unsigned long long* __restrict__ cand, TopkState* st, int n,
                                                           int k, int64_t* out_row, int64_t* out_col, float* out_val,
                                                           int32_t* out_count) { topk_final_kernel_body(blockIdx, gridDim, cand, st, n, k, out_row, out_col, out_val, out_count); }


}  // namespace

namespace {
// global top-k of the [M, N] score matrix (the multi-workgroup path + its gated one-workgroup fallback)
void launch_topk(const float* scores, int M, int N, int LD, int k, unsigned* ghist, unsigned long long* cand, int64_t* ref_idx,
                 int64_t* src_idx, float* out_scores, int32_t* out_count, hipStream_t st) {
  using namespace rdm;
  TopkState* ts = reinterpret_cast<TopkState*>(ghist + 2 * kTopkBins);
  const int tb = ceil_div(M, kTopkRows);
  fill_words<unsigned>(ghist, 2 * kTopkBins + 8, 0u, st);
  ::rdm::launch<topk_hist_kernel_body<0>, topk_hist_kernel<0>, 256>(dim3(tb), 0, st, scores, M, N, LD, ts, ghist);
  ::rdm::launch<topk_pick_kernel_body<0>, topk_pick_kernel<0>, 64>(dim3(1), 0, st, ghist, k, ts);
  ::rdm::launch<topk_hist_kernel_body<1>, topk_hist_kernel<1>, 256>(dim3(tb), 0, st, scores, M, N, LD, ts, ghist + kTopkBins);
  ::rdm::launch<topk_pick_kernel_body<1>, topk_pick_kernel<1>, 64>(dim3(1), 0, st, ghist + kTopkBins, k, ts);
  ::rdm::launch<topk_collect_kernel_body, topk_collect_kernel, 256>(dim3(tb), 0, st, scores, M, N, LD, ts, cand);
  ::rdm::launch<topk_final_kernel_body, topk_final_kernel, 1024>(dim3(1), 0, st, cand, ts, N, k, ref_idx, src_idx, out_scores, out_count);
  ::rdm::launch<topk_kernel_body, topk_kernel, 1024>(dim3(1), 0, st, scores, M, N, LD, k, ref_idx, src_idx, out_scores,
                     out_count, reinterpret_cast<const unsigned*>(ts));  // only runs if the candidate list overflowed
}
}  // namespace

extern "C" size_t rdm_coarse_matching_workspace_bytes(int64_t m, int64_t n) {
  rdm::Arena a(nullptr, 0);
  a.take<float>(m > 0 ? m : 1);
  a.take<float>(n > 0 ? n : 1);
  a.take<unsigned>(2 * kTopkBins + 8);
  a.take<unsigned long long>(kTopkCand);
  return a.off;
}

// scores: in = f_ref . f_src^T  [m, n] (ld), overwritten with the dual-normalised matching scores.
extern "C" int rdm_coarse_matching(float* scores, int64_t m, int64_t n, int64_t ld, const uint8_t* ref_mask,
                                   const uint8_t* src_mask, int dual_normalization, int k, int64_t* ref_idx,
                                   int64_t* src_idx, float* out_scores, int32_t* out_count, void* ws,
                                   size_t ws_bytes, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(scores && ref_mask && src_mask && ref_idx && src_idx && out_scores && out_count,
              "rdm_coarse_matching: null pointer");
  RDM_REQUIRE(m > 0 && n > 0 && k > 0 && k <= 1024 && m * n < (1ll << 31), "rdm_coarse_matching: bad sizes");
  Arena ar(ws, ws_bytes);
  float* rsum = ar.take<float>(m);
  float* csum = ar.take<float>(n);
  unsigned* ghist = ar.take<unsigned>(2 * kTopkBins + 8);  // two histograms + TopkState behind them
  unsigned long long* cand = ar.take<unsigned long long>(kTopkCand);
  if (!ar.ok) {
    set_error("rdm_coarse_matching: workspace too small");
    return RDM_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int M = static_cast<int>(m), N = static_cast<int>(n), LD = static_cast<int>(ld);
  const int eb = static_cast<int>(ceil_div<int64_t>(m * n, 256));
  ::rdm::launch<coarse_scores_kernel_body, coarse_scores_kernel, 256>(dim3(eb), 0, st, scores, M, N, LD, ref_mask, src_mask);
  if (dual_normalization) {
    ::rdm::launch<coarse_rowsum_kernel_body, coarse_rowsum_kernel, 256>(dim3(ceil_div(M, 4)), 0, st, scores, M, N, LD, rsum);
    ::rdm::launch<coarse_colsum_kernel_body, coarse_colsum_kernel, 256>(dim3(ceil_div(N, 256)), 0, st, scores, M, N, LD, csum);
  }
  ::rdm::launch<coarse_dual_kernel_body, coarse_dual_kernel, 256>(dim3(eb), 0, st, scores, M, N, LD,
                     dual_normalization ? rsum : static_cast<const float*>(nullptr),
                     dual_normalization ? csum : static_cast<const float*>(nullptr), ref_mask, src_mask);
  launch_topk(scores, M, N, LD, k, ghist, cand, ref_idx, src_idx, out_scores, out_count, st);
  return launch_status("coarse matching kernels");
}

extern "C" size_t rdm_coarse_matching_features_workspace_bytes(int64_t m, int64_t n) {
  rdm::Arena a(nullptr, 0);
  const int64_t ld = (n + 3) / 4 * 4;
  a.take<double>(static_cast<size_t>(m > 0 ? m : 1) * ld);
  a.take<float>(static_cast<size_t>(m > 0 ? m : 1) * ld);
  a.take<double>(m > 0 ? m : 1);
  a.take<double>(n > 0 ? n : 1);
  a.take<unsigned>(2 * kTopkBins + 8);
  a.take<unsigned long long>(kTopkCand);
  return a.off;
}

extern "C" int rdm_coarse_matching_features(const float* ref_feats, int64_t ld_ref, int64_t m, const float* src_feats,
                                            int64_t ld_src, int64_t n, int64_t d, const uint8_t* ref_mask,
                                            const uint8_t* src_mask, int dual_normalization, int k, int64_t* ref_idx,
                                            int64_t* src_idx, float* out_scores, int32_t* out_count, void* ws,
                                            size_t ws_bytes, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(ref_feats && src_feats && ref_mask && src_mask && ref_idx && src_idx && out_scores && out_count,
              "rdm_coarse_matching_features: null pointer");
  RDM_REQUIRE(m > 0 && n > 0 && d > 0 && d <= 448 && k > 0 && k <= 1024 && m * n < (1ll << 31),
              "rdm_coarse_matching_features: bad sizes (feature width <= 448: two 16-row tiles in 64 KB of LDS)");
  Arena ar(ws, ws_bytes);
  const int64_t ld = (n + 3) / 4 * 4;
  double* s64 = ar.take<double>(static_cast<size_t>(m) * ld);
  float* s32 = ar.take<float>(static_cast<size_t>(m) * ld);
  double* rsum = ar.take<double>(m);
  double* csum = ar.take<double>(n);
  unsigned* ghist = ar.take<unsigned>(2 * kTopkBins + 8);
  unsigned long long* cand = ar.take<unsigned long long>(kTopkCand);
  if (!ar.ok) {
    set_error("rdm_coarse_matching_features: workspace too small");
    return RDM_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int M = static_cast<int>(m), N = static_cast<int>(n), LD = static_cast<int>(ld), D = static_cast<int>(d);
  const size_t lds = sizeof(float) * 2 * 16 * (D + 1);
  ::rdm::launch<coarse_scores64_kernel_body, coarse_scores64_kernel, 256>(dim3(ceil_div(N, 16), ceil_div(M, 16)), lds, st, ref_feats,
                     static_cast<int>(ld_ref), M, src_feats, static_cast<int>(ld_src), N, D, ref_mask, src_mask, s64, LD);
  if (dual_normalization) {
    ::rdm::launch<coarse_rowsum64_kernel_body, coarse_rowsum64_kernel, 256>(dim3(ceil_div(M, 4)), 0, st, s64, M, N, LD, rsum);
    ::rdm::launch<coarse_colsum64_kernel_body, coarse_colsum64_kernel, 256>(dim3(ceil_div(N, 64)), 0, st, s64, M, N, LD, csum);
  }
  ::rdm::launch<coarse_dual64_kernel_body, coarse_dual64_kernel, 256>(dim3(static_cast<unsigned>(ceil_div<int64_t>(m * n, 256))), 0, st, s64, M, N, LD,
                     dual_normalization ? rsum : static_cast<const double*>(nullptr),
                     dual_normalization ? csum : static_cast<const double*>(nullptr), ref_mask, src_mask, s32, LD);
  launch_topk(s32, M, N, LD, k, ghist, cand, ref_idx, src_idx, out_scores, out_count, st);
  return launch_status("coarse matching (features) kernels");
}
