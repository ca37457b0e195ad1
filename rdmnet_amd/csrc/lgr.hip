// a15/a16 -- dense correspondence extraction and local-to-global registration, entirely on the GPU.
//
// Reference: geotransformer/modules/geotransformer/local_global_registration.py:49-91
// (compute_correspondence_matrix, k = 1, dustbin, non-mutual), :145-202 (local_to_global_registration),
// :93-136 (convert_to_batch), geotransformer/modules/registration/procrustes.py:6-73 (weighted
// Kabsch; the reference moves H to the CPU for torch.svd -- here nothing leaves the device).
//
// Stages (one launch each, sizes stay on the device):
//   extract   per patch: S = exp(log scores); row/column top-1 incl. dustbin; keep (i,j) iff it beats
//             the dustbin from either side and both points are valid; list them row-major (the order
//             of torch.nonzero)                                                            :204-243
//   layout    exclusive scan of the per-patch counts, chunk list = patches with >= 3 matches
//   gather    stacked correspondences (ref point, src point, score) in patch order
//   local     per chunk: weighted Procrustes                                               :175-181
//   score     per chunk: #correspondences (of ALL) with residual < acceptance radius       :182-187
//   refine    first argmax, then 1 + (steps-1) global Procrustes rounds                    :188-200
// The rotation is obtained with Horn's quaternion form of the Kabsch problem (largest eigenvector of
// a symmetric 4x4, Jacobi in fp64): identical to V diag(1,1,det) U^T wherever that is unique, and
// the identity for a zero covariance like torch.svd.  Sums are accumulated in fp64.
#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "procrustes.h"
#include "lockstep.h"

namespace {

using namespace rdm;

constexpr int kSide = 128;           // points per patch
constexpr int kPerPatch = 2 * kSide; // upper bound of matches per patch (one per row + one per column)

struct LgrBuffers {
  int32_t* patch_count;   // [B]
  int32_t* patch_offset;  // [B]
  int32_t* chunk_patch;   // [B]
  int32_t* meta;          // [0] = C, [1] = number of chunks, [2] = best chunk
  int32_t* local_i;       // [B, kPerPatch]
  int32_t* local_j;       // [B, kPerPatch]
  float* local_s;         // [B, kPerPatch]
  float* chunk_T;         // [B, 12]
  int32_t* chunk_inliers; // [B]
};

// ------------------------------------------------------------------------------------------ extract
// Only the block of REAL rows and columns (+ the dustbin line of each side) is staged: a masked line of the Sinkhorn output
// holds fl(-1e12), i.e. exp() = 0 exactly, while a real row (column) of exp(scores) sums to 1, so its maximum is positive
// and a masked entry never wins nor ties a top-1 -- the compacted search returns the indices of the full 129 x 129 one.  A
// patch holds a few dozen real points of 128: 1-2 k exponentials per patch instead of 16.6 k.
__device__ __forceinline__ void lgr_extract_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* log_scores, int side,
                                                          const unsigned char* ref_mask,
                                                          const unsigned char* src_mask, LgrBuffers w) {
  (void)blockIdx; (void)gridDim;
  extern __shared__ float S[];  // compacted [(nr+1)][(nc+1)], row stride ldc (odd)
  __shared__ int rows[kSide + 1], cols[kSide + 1];  // original index of every compacted line, dustbin (= side) last
  __shared__ int rowarg[kSide + 1], colarg[kSide + 1];
  __shared__ unsigned char rowok[kSide + 1], colok[kSide + 1];
  __shared__ int rowcnt[kSide + 1];
  __shared__ int s_n[2];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n1 = side + 1;
  const float* L = log_scores + static_cast<int64_t>(b) * n1 * n1;
  if (wave < 2) {  // wavefront 0: the real rows, wavefront 1: the real columns, ascending (ballot compaction)
    const unsigned char* mk = (wave == 0 ? ref_mask : src_mask) + static_cast<int64_t>(b) * side;
    int* list = wave == 0 ? rows : cols;
    int cnt = 0;
    for (int i0 = 0; i0 < side; i0 += 64) {
      const int i = i0 + lane;
      const bool on = i < side && mk[i] != 0;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(on);
      if (on) list[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = i;
      cnt += __popcll(bal);
    }
    if (lane == 0) {
      list[cnt] = side;
      s_n[wave] = cnt;
    }
  }
  __syncthreads();
  const int nr = s_n[0], nc = s_n[1], R = nr + 1, C = nc + 1, ldc = C | 1;
  for (int t = tid; t < R * C; t += 256) {
    const int r = t / C, c = t % C;
    S[r * ldc + c] = expf(L[static_cast<int64_t>(rows[r]) * n1 + cols[c]]);
  }
  __syncthreads();
  if (tid < nr) {  // top-1 of a real row over the real columns + dustbin (first maximum), must beat the dustbin column
    const int r = tid;
    float best = S[r * ldc];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      const float v = S[r * ldc + c];
      if (v > best) {
        best = v;
        arg = c;
      }
    }
    rowarg[r] = arg;  // compacted column (nc = dustbin)
    rowok[r] = (arg < nc && best > S[r * ldc + nc]) ? 1 : 0;
  } else if (tid >= 128 && tid < 128 + nc) {  // top-1 of a real column over the real rows + dustbin
    const int c = tid - 128;
    float best = S[c];
    int arg = 0;
    for (int r = 1; r < R; ++r) {
      const float v = S[r * ldc + c];
      if (v > best) {
        best = v;
        arg = r;
      }
    }
    colarg[c] = arg;
    colok[c] = (arg < nr && best > S[nr * ldc + c]) ? 1 : 0;
  }
  __syncthreads();
  auto is_corr = [&](int r, int c) { return (rowok[r] && rowarg[r] == c) || (colok[c] && colarg[c] == r); };
  if (tid < nr) {
    int cn = 0;
    for (int c = 0; c < nc; ++c) cn += is_corr(tid, c) ? 1 : 0;
    rowcnt[tid] = cn;
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int r = 0; r < nr; ++r) {
      const int cn = rowcnt[r];
      rowcnt[r] = acc;
      acc += cn;
    }
    w.patch_count[b] = acc;
  }
  __syncthreads();
  if (tid < nr) {  // row-major over the original indices (the order of torch.nonzero): the lists are ascending
    int pos = rowcnt[tid];
    for (int c = 0; c < nc; ++c)
      if (is_corr(tid, c)) {
        const int64_t o = static_cast<int64_t>(b) * kPerPatch + pos++;
        w.local_i[o] = rows[tid];
        w.local_j[o] = cols[c];
        w.local_s[o] = S[tid * ldc + c];
      }
  }
}
__global__ __launch_bounds__(256) void lgr_extract_kernel(const float* log_scores, int side,
                                                          const unsigned char* ref_mask,
                                                          const unsigned char* src_mask, LgrBuffers w) { lgr_extract_kernel_body(blockIdx, gridDim, log_scores, side, ref_mask, src_mask, w); }


// ------------------------------------------------------------------------------------------ layout
// One wavefront: exclusive prefix of the per-patch correspondence counts (= offsets in nonzero order) and the
// ordered list of patches with >= min_corr correspondences.  Lane l owns patches 4l .. 4l+3 of each 256-patch chunk.
__device__ __forceinline__ void lgr_layout_kernel_body(const dim3 blockIdx, const dim3 gridDim, int batch, int min_corr, LgrBuffers w) {
  (void)blockIdx; (void)gridDim;
  const int lane = threadIdx.x;
  int acc = 0, chunks = 0;  // running totals, identical in all lanes
  for (int base = 0; base < batch; base += 256) {
    int c[4], local = 0, nbig = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = base + 4 * lane + u;
      c[u] = b < batch ? w.patch_count[b] : 0;
      local += c[u];
      nbig += (b < batch && c[u] >= min_corr) ? 1 : 0;
    }
    int inc = local, inb = nbig;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64), tb = __shfl_up(inb, o, 64);
      if (lane >= o) {
        inc += t;
        inb += tb;
      }
    }
    int off = acc + inc - local, slot = chunks + inb - nbig;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = base + 4 * lane + u;
      if (b < batch) {
        w.patch_offset[b] = off;
        off += c[u];
        if (c[u] >= min_corr) w.chunk_patch[slot++] = b;
      }
    }
    acc += __shfl(inc, 63, 64);
    chunks += __shfl(inb, 63, 64);
  }
  if (lane == 0) {
    w.meta[0] = acc;
    w.meta[1] = chunks;
    w.meta[2] = -1;
  }
}
__global__ __launch_bounds__(64) void lgr_layout_kernel(int batch, int min_corr, LgrBuffers w) { lgr_layout_kernel_body(blockIdx, gridDim, batch, min_corr, w); }


__device__ __forceinline__ void lgr_gather_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* ref_knn, const float* src_knn, int side, LgrBuffers w,
                                  float* ref_corr, float* src_corr, float* corr_scores) {
  (void)blockIdx; (void)gridDim;
  const int b = blockIdx.x;
  const int cnt = w.patch_count[b], off = w.patch_offset[b];
  for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
    const int64_t o = static_cast<int64_t>(b) * kPerPatch + t;
    const float* r = ref_knn + (static_cast<int64_t>(b) * side + w.local_i[o]) * 3;
    const float* s = src_knn + (static_cast<int64_t>(b) * side + w.local_j[o]) * 3;
    for (int d = 0; d < 3; ++d) {
      ref_corr[3 * (off + t) + d] = r[d];
      src_corr[3 * (off + t) + d] = s[d];
    }
    corr_scores[off + t] = w.local_s[o];
  }
}
__global__ void lgr_gather_kernel(const float* ref_knn, const float* src_knn, int side, LgrBuffers w,
                                  float* ref_corr, float* src_corr, float* corr_scores) { lgr_gather_kernel_body(blockIdx, gridDim, ref_knn, src_knn, side, w, ref_corr, src_corr, corr_scores); }


// ------------------------------------------------------------------------------------------ Procrustes
// Sum N values per thread over the block (fixed order); red must hold N * 16 doubles.
template <int N>
__device__ void block_sum_n(double (&v)[N], double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < N; ++k) red[k * 16 + wv] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += red[k * 16 + i];
    v[k] = t;
  }
}

// Weighted rigid fit src -> ref over entries [x0, x1) (procrustes.py:36-73).  Every thread of the
// block must call it; T (3x4, row-major R|t) is valid in all threads on return.
__device__ void block_procrustes(const float* src, const float* ref, const float* wts, const unsigned char* gate,
                                 int x0, int x1, double* red, double T[12], double (*basis)[4] = nullptr) {
  double sw = 0.0;
  for (int i = x0 + threadIdx.x; i < x1; i += blockDim.x) {
    float w = wts[i];
    if (gate && !gate[i]) w = 0.f;
    if (w < 0.f) w = 0.f;  // weight_thresh = 0
    sw += w;
  }
  {
    double t[1] = {sw};
    block_sum_n<1>(t, red);
    sw = t[0] + 1e-5;
  }
  const double inv_sw = 1.0 / sw;  // one fp64 division per fit instead of one per correspondence and pass
  double cs[3] = {0, 0, 0}, cr[3] = {0, 0, 0};
  for (int i = x0 + threadIdx.x; i < x1; i += blockDim.x) {
    float w = wts[i];
    if ((gate && !gate[i]) || w < 0.f) w = 0.f;
    const double wn = w * inv_sw;
    for (int d = 0; d < 3; ++d) {
      cs[d] += wn * src[3 * i + d];
      cr[d] += wn * ref[3 * i + d];
    }
  }
  {
    double t[6] = {cs[0], cs[1], cs[2], cr[0], cr[1], cr[2]};
    block_sum_n<6>(t, red);
    for (int d = 0; d < 3; ++d) {
      cs[d] = t[d];
      cr[d] = t[3 + d];
    }
  }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // H[a][b] = sum w (s_a - cs_a)(r_b - cr_b)
  for (int i = x0 + threadIdx.x; i < x1; i += blockDim.x) {
    float w = wts[i];
    if ((gate && !gate[i]) || w < 0.f) w = 0.f;
    const double wn = w * inv_sw;
    double s[3], r[3];
    for (int d = 0; d < 3; ++d) {
      s[d] = src[3 * i + d] - cs[d];
      r[d] = ref[3 * i + d] - cr[d];
    }
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) H[3 * a + b] += wn * s[a] * r[b];
  }
  block_sum_n<9>(H, red);
  const double Sxx = H[0], Sxy = H[1], Sxz = H[2], Syx = H[3], Syy = H[4], Syz = H[5], Szx = H[6], Szy = H[7],
               Szz = H[8];
  double N[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                    {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                    {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                    {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
  double q[4];
  horn_quaternion(N, q, basis);
  const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw),     2 * (qx * qz + qy * qw),
                 2 * (qx * qy + qz * qw),     1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                 2 * (qx * qz - qy * qw),     2 * (qy * qz + qx * qw),     1 - 2 * (qx * qx + qy * qy)};
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) T[4 * a + b] = R[3 * a + b];
    T[4 * a + 3] = cr[a] - (R[3 * a] * cs[0] + R[3 * a + 1] * cs[1] + R[3 * a + 2] * cs[2]);
  }
}

// residual < radius in the reference's fp32 arithmetic (apply_transform + linalg.norm)
__device__ __forceinline__ bool is_inlier(const float* T, const float* src, const float* ref, int i, float radius) {
  const float x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
  const float ax = fmaf(z, T[2], fmaf(y, T[1], x * T[0])) + T[3];
  const float ay = fmaf(z, T[6], fmaf(y, T[5], x * T[4])) + T[7];
  const float az = fmaf(z, T[10], fmaf(y, T[9], x * T[8])) + T[11];
  const float dx = ref[3 * i] - ax, dy = ref[3 * i + 1] - ay, dz = ref[3 * i + 2] - az;
  return __fsqrt_rn((dx * dx + dy * dy) + dz * dz) < radius;
}

// one 64-thread block per chunk: local Procrustes, then inlier count over ALL correspondences
__device__ __forceinline__ void lgr_local_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* ref_corr, const float* src_corr,
                                                        const float* scores, float radius, LgrBuffers w) {
  (void)blockIdx; (void)gridDim;
  __shared__ double red[9 * 16];
  __shared__ float Tf[12];
  __shared__ int cnt_sh;
  const int chunk = blockIdx.x;
  if (chunk >= w.meta[1]) return;
  const int b = w.chunk_patch[chunk];
  const int x0 = w.patch_offset[b], x1 = x0 + w.patch_count[b];
  double T[12];
  block_procrustes(src_corr, ref_corr, scores, nullptr, x0, x1, red, T);
  if (threadIdx.x < 12) {
    Tf[threadIdx.x] = static_cast<float>(T[threadIdx.x]);
    w.chunk_T[12 * chunk + threadIdx.x] = Tf[threadIdx.x];
  }
  if (threadIdx.x == 0) cnt_sh = 0;
  __syncthreads();
  const int C = w.meta[0];
  int c = 0;
  for (int i = threadIdx.x; i < C; i += blockDim.x) c += is_inlier(Tf, src_corr, ref_corr, i, radius) ? 1 : 0;
  c = wave_sum_i(c);
  if ((threadIdx.x & 63) == 0) atomicAdd(&cnt_sh, c);
  __syncthreads();
  if (threadIdx.x == 0) w.chunk_inliers[chunk] = cnt_sh;
}
__global__ __launch_bounds__(256) void lgr_local_kernel(const float* ref_corr, const float* src_corr,
                                                        const float* scores, float radius, LgrBuffers w) { lgr_local_kernel_body(blockIdx, gridDim, ref_corr, src_corr, scores, radius, w); }


// single block: pick the hypothesis, refine globally
constexpr int kRefineStage = 4608;  // correspondences staged in LDS: 4608 * 29 B = 131 KB
__device__ __forceinline__ void lgr_refine_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* ref_corr, const float* src_corr,
                                                          const float* scores, float radius, int steps,
                                                          LgrBuffers w, unsigned char* gate, float* out_T,
                                                          int32_t* counts) {
  (void)blockIdx; (void)gridDim;
  __shared__ double red[9 * 16];
  __shared__ float Tf[12];
  extern __shared__ float stage[];
  const int C = w.meta[0], chunks = w.meta[1];
  int best_out = w.meta[2];  // (the layout kernel's value when no patch reaches the threshold)
  // The refinement makes 4 passes over the correspondences per step; one workgroup, so every pass is a chain of
  // dependent L2 round trips.  Up to kRefineStage correspondences are copied into LDS once (7 words each + the gate).
  if (C <= kRefineStage) {
    float* ls = stage;
    float* lr = ls + 3 * C;
    float* lw = lr + 3 * C;
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) {
      ls[i] = src_corr[i];
      lr[i] = ref_corr[i];
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x) lw[i] = scores[i];
    src_corr = ls; ref_corr = lr; scores = lw;
    gate = reinterpret_cast<unsigned char*>(lw + C);
    __syncthreads();
  }
  double T[12];
  if (chunks > 0) {
    // first maximum of the inlier counts (torch.argmax): block-wide max of (count, -index) packed in 64 bits
    __shared__ unsigned long long s_best[4];
    unsigned long long mine = 0;
    for (int k = threadIdx.x; k < chunks; k += blockDim.x)
      mine = max(mine, (static_cast<unsigned long long>(static_cast<unsigned>(w.chunk_inliers[k])) << 32) |
                           static_cast<unsigned>(0x7fffffff - k));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine = max(mine, static_cast<unsigned long long>(__shfl_xor(mine, o, 64)));
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = mine;
    __syncthreads();
    const unsigned long long all = max(max(s_best[0], s_best[1]), max(s_best[2], s_best[3]));
    const int best = 0x7fffffff - static_cast<int>(all & 0xffffffffull);
    if (threadIdx.x == 0) w.meta[2] = best;
    best_out = best;
    if (threadIdx.x < 12) Tf[threadIdx.x] = w.chunk_T[12 * best + threadIdx.x];
  } else {  // degenerate: no patch reaches the threshold -> start from all correspondences (:189-194)
    block_procrustes(src_corr, ref_corr, scores, nullptr, 0, C, red, T);
    if (threadIdx.x < 12) Tf[threadIdx.x] = static_cast<float>(T[threadIdx.x]);
  }
  __syncthreads();
  double basis[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};  // eigenvector frame carried from step to step
  for (int step = 0; step < steps; ++step) {
    for (int i = threadIdx.x; i < C; i += blockDim.x) gate[i] = is_inlier(Tf, src_corr, ref_corr, i, radius) ? 1 : 0;
    __syncthreads();
    block_procrustes(src_corr, ref_corr, scores, gate, 0, C, red, T, basis);
    __syncthreads();
    if (threadIdx.x < 12) Tf[threadIdx.x] = static_cast<float>(T[threadIdx.x]);
    __syncthreads();
  }
  if (threadIdx.x < 16) {
    const int r = threadIdx.x / 4, c = threadIdx.x % 4;
    out_T[threadIdx.x] = r < 3 ? Tf[4 * r + c] : (c == 3 ? 1.f : 0.f);
  }
  if (threadIdx.x == 0) {  // {n_correspondences, n_hypotheses, best_hypothesis} for the caller (no separate copy launch)
    counts[0] = C;
    counts[1] = chunks;
    counts[2] = best_out;
  }
}
__global__ __launch_bounds__(256) void lgr_refine_kernel(const float* ref_corr, const float* src_corr,
                                                          const float* scores, float radius, int steps,
                                                          LgrBuffers w, unsigned char* gate, float* out_T,
                                                          int32_t* counts) { lgr_refine_kernel_body(blockIdx, gridDim, ref_corr, src_corr, scores, radius, steps, w, gate, out_T, counts); }


}  // namespace

extern "C" size_t rdm_lgr_workspace_bytes(int64_t batch) {
  rdm::Arena a(nullptr, 0);
  const size_t b = static_cast<size_t>(batch > 0 ? batch : 1);
  a.take<int32_t>(b); a.take<int32_t>(b); a.take<int32_t>(b); a.take<int32_t>(4);
  a.take<int32_t>(b * kPerPatch); a.take<int32_t>(b * kPerPatch); a.take<float>(b * kPerPatch);
  a.take<float>(b * 12); a.take<int32_t>(b);
  a.take<unsigned char>(b * kPerPatch);
  return a.off;
}

// log_scores [batch, side+1, side+1] (Sinkhorn output), knn points [batch, side, 3], masks [batch, side].
// Outputs (capacity batch*2*side rows): ref_corr/src_corr [.,3], corr_scores [.], transform [16],
// counts [3] = {n_correspondences, n_hypotheses, best_hypothesis} (device int32).
extern "C" int rdm_lgr(const float* log_scores, const float* ref_knn_points, const float* src_knn_points,
                       const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks, int64_t batch, int64_t side,
                       float acceptance_radius, int correspondence_threshold, int num_refinement_steps,
                       float* ref_corr, float* src_corr, float* corr_scores, float* transform, int32_t* counts,
                       void* ws, size_t ws_bytes, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(log_scores && ref_knn_points && src_knn_points && ref_knn_masks && src_knn_masks && ref_corr &&
                  src_corr && corr_scores && transform && counts, "rdm_lgr: null pointer");
  RDM_REQUIRE(batch > 0 && side > 0 && side <= kSide, "rdm_lgr: bad sizes (batch=%lld side=%lld)",
              (long long)batch, (long long)side);
  Arena ar(ws, ws_bytes);
  const size_t b = static_cast<size_t>(batch);
  LgrBuffers w;
  w.patch_count = ar.take<int32_t>(b);
  w.patch_offset = ar.take<int32_t>(b);
  w.chunk_patch = ar.take<int32_t>(b);
  w.meta = ar.take<int32_t>(4);
  w.local_i = ar.take<int32_t>(b * kPerPatch);
  w.local_j = ar.take<int32_t>(b * kPerPatch);
  w.local_s = ar.take<float>(b * kPerPatch);
  w.chunk_T = ar.take<float>(b * 12);
  w.chunk_inliers = ar.take<int32_t>(b);
  unsigned char* gate = ar.take<unsigned char>(b * kPerPatch);
  if (!ar.ok) {
    set_error("rdm_lgr: workspace too small (%zu < %zu)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int S = static_cast<int>(side), B = static_cast<int>(batch);
  const size_t lds = sizeof(float) * (S + 1) * ((S + 1) | 1);
  static std::atomic<uint64_t> attr_extract{0}, attr_refine{0};  // per device (rdm::set_max_dynamic_lds)
  RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(lgr_extract_kernel), 160 * 1024 - 4096, attr_extract));
  RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(lgr_refine_kernel), 160 * 1024 - 4096, attr_refine));
  ::rdm::launch<lgr_extract_kernel_body, lgr_extract_kernel, 256>(dim3(B), lds, st, log_scores, S, ref_knn_masks, src_knn_masks, w);
  ::rdm::launch<lgr_layout_kernel_body, lgr_layout_kernel, 64>(dim3(1), 0, st, B, correspondence_threshold, w);
  ::rdm::launch<lgr_gather_kernel_body, lgr_gather_kernel, 64>(dim3(B), 0, st, ref_knn_points, src_knn_points, S, w, ref_corr,
                     src_corr, corr_scores);
  ::rdm::launch<lgr_local_kernel_body, lgr_local_kernel, 256>(dim3(B), 0, st, ref_corr, src_corr, corr_scores, acceptance_radius, w);
  ::rdm::launch<lgr_refine_kernel_body, lgr_refine_kernel, 256>(dim3(1), kRefineStage * 29 + 64, st, ref_corr, src_corr, corr_scores,
                     acceptance_radius, num_refinement_steps, w, gate, transform, counts);
  return launch_status("lgr kernels");
}
