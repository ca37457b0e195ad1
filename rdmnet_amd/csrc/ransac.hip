// SURVEY.md §8f rank 4 -- RANSAC pose from the path's correspondences, the reference's second evaluation mode
// (experiments/infer.py:75-82, experiments/eval.py:179-186 -> geotransformer/utils/open3d.py:173-203:
// Open3D registration_ransac_based_on_correspondence, point-to-point, ransac_n = 4, 50 000 iterations,
// distance threshold 0.3 m).
//
// Open3D (0.11.2) is not under /root/reference -> PARITY UNPINNED.  Restated algorithm (Open3D's
// RegistrationRANSACBasedOnCorrespondence): every iteration draws ransac_n correspondences uniformly WITH
// replacement, fits a rigid transform to them (Umeyama without scale = Kabsch), counts the correspondences
// whose transformed source lies closer than the threshold to its target (fitness) and their RMSE; the best
// iteration wins by (fitness, then lower RMSE, then first); its transform is returned as is (no refit).
// Open3D draws from a global Mersenne twister; here draw j of iteration i is a counter-based hash of
// (seed, i, j) so that all iterations are independent -- 50 000 hypotheses are scored in parallel, one
// wavefront each -- and the result is a deterministic function of (inputs, seed); oracle/preprocess.py
// restates exactly this.
#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "procrustes.h"

#pragma clang fp contract(off)

namespace {
using namespace rdm;

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// one thread per iteration: sample, fit (float64), store R|t as 12 doubles
__global__ __launch_bounds__(256) void ransac_fit_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                          int n_corr, int ransac_n, int iters, unsigned long long seed,
                                                          double* __restrict__ hyp) {
  const int it = blockIdx.x * 256 + threadIdx.x;
  if (it >= iters) return;
  double s[8][3], r[8][3], cs[3] = {0, 0, 0}, cr[3] = {0, 0, 0};
  for (int j = 0; j < ransac_n; ++j) {
    const unsigned long long z = mix64(seed + 0x9E3779B97F4A7C15ull * (static_cast<unsigned long long>(it) * ransac_n + j + 1));
    const int c = static_cast<int>(z % static_cast<unsigned long long>(n_corr));
    for (int d = 0; d < 3; ++d) {
      s[j][d] = src[3 * c + d];
      r[j][d] = ref[3 * c + d];
      cs[d] += s[j][d];
      cr[d] += r[j][d];
    }
  }
  for (int d = 0; d < 3; ++d) {
    cs[d] /= ransac_n;
    cr[d] /= ransac_n;
  }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < ransac_n; ++j)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) H[3 * a + b] += (s[j][a] - cs[a]) * (r[j][b] - cr[b]);
  double R[9];
  kabsch_rotation(H, R);
  double* T = hyp + 12ll * it;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) T[4 * a + b] = R[3 * a + b];
    T[4 * a + 3] = cr[a] - (R[3 * a] * cs[0] + R[3 * a + 1] * cs[1] + R[3 * a + 2] * cs[2]);
  }
}

// one wavefront per iteration: inlier count and squared error over all correspondences (float64, fixed order)
__global__ __launch_bounds__(256) void ransac_score_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                            int n_corr, double threshold, int iters,
                                                            const double* __restrict__ hyp, int32_t* __restrict__ inliers,
                                                            double* __restrict__ err2) {
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (it >= iters) return;
  double T[12];
  for (int k = 0; k < 12; ++k) T[k] = hyp[12ll * it + k];
  int cnt = 0;
  double e2 = 0.0;
  for (int c = lane; c < n_corr; c += 64) {
    const double x = src[3 * c], y = src[3 * c + 1], z = src[3 * c + 2];
    const double dx = (T[0] * x + T[1] * y + T[2] * z + T[3]) - ref[3 * c];
    const double dy = (T[4] * x + T[5] * y + T[6] * z + T[7]) - ref[3 * c + 1];
    const double dz = (T[8] * x + T[9] * y + T[10] * z + T[11]) - ref[3 * c + 2];
    const double d2 = dx * dx + dy * dy + dz * dz;
    if (sqrt(d2) < threshold) {
      ++cnt;
      e2 += d2;
    }
  }
  cnt = wave_sum_i(cnt);
  e2 = wave_sum(e2);
  if (lane == 0) {
    inliers[it] = cnt;
    err2[it] = e2;
  }
}

// best iteration: more inliers, then lower RMSE, then lower index (Open3D keeps the first of equals)
__global__ __launch_bounds__(1024) void ransac_best_kernel(int iters, const double* __restrict__ hyp,
                                                            const int32_t* __restrict__ inliers, const double* __restrict__ err2,
                                                            float* __restrict__ transform, int32_t* __restrict__ stats,
                                                            float* __restrict__ rmse_out) {
  __shared__ int s_cnt[1024], s_it[1024];
  __shared__ double s_rmse[1024];
  int bc = -1, bi = 0x7fffffff;
  double br = 0.0;
  for (int it = threadIdx.x; it < iters; it += 1024) {
    const int c = inliers[it];
    const double r = c > 0 ? sqrt(err2[it] / c) : 0.0;
    if (c > bc || (c == bc && r < br)) {  // ascending `it` per thread: ties keep the earlier one
      bc = c; br = r; bi = it;
    }
  }
  s_cnt[threadIdx.x] = bc; s_rmse[threadIdx.x] = br; s_it[threadIdx.x] = bi;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const int c2 = s_cnt[threadIdx.x + o], i2 = s_it[threadIdx.x + o];
      const double r2 = s_rmse[threadIdx.x + o];
      const int c1 = s_cnt[threadIdx.x], i1 = s_it[threadIdx.x];
      const double r1 = s_rmse[threadIdx.x];
      const bool take = c2 > c1 || (c2 == c1 && (r2 < r1 || (r2 == r1 && i2 < i1)));
      if (take) {
        s_cnt[threadIdx.x] = c2; s_rmse[threadIdx.x] = r2; s_it[threadIdx.x] = i2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 16) {
    const int best = s_it[0];
    float v = (threadIdx.x == 15) ? 1.f : 0.f;
    if (threadIdx.x < 12 && s_cnt[0] > 0) v = static_cast<float>(hyp[12ll * best + threadIdx.x]);
    else if (threadIdx.x < 12) v = (threadIdx.x % 5 == 0) ? 1.f : 0.f;  // no inlier at all: identity, as Open3D's empty result
    transform[threadIdx.x] = v;
  }
  if (threadIdx.x == 0) {
    stats[0] = s_cnt[0] > 0 ? s_it[0] : -1;
    stats[1] = s_cnt[0] > 0 ? s_cnt[0] : 0;
    *rmse_out = static_cast<float>(s_rmse[0]);
  }
}

}  // namespace

extern "C" size_t rdm_ransac_workspace_bytes(int num_iterations) {
  const size_t n = static_cast<size_t>(num_iterations > 0 ? num_iterations : 1);
  return rdm::align_up(n * 12 * sizeof(double)) + rdm::align_up(n * sizeof(double)) + rdm::align_up(n * sizeof(int32_t));
}

extern "C" int rdm_ransac_correspondences(const float* src_corr, const float* ref_corr, int64_t n_corr, float distance_threshold,
                                          int ransac_n, int num_iterations, uint64_t seed, float* transform, int32_t* stats,
                                          float* inlier_rmse, int32_t* hyp_inliers, void* ws, size_t ws_bytes, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(transform && stats && inlier_rmse, "rdm_ransac_correspondences: null output");
  RDM_REQUIRE(n_corr >= 0 && n_corr < (1ll << 31) && ransac_n >= 3 && ransac_n <= 8 && num_iterations > 0 &&
                  distance_threshold > 0.f,
              "rdm_ransac_correspondences: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  Arena ar(ws, ws_bytes);
  double* hyp = ar.take<double>(static_cast<size_t>(num_iterations) * 12);
  double* err2 = ar.take<double>(num_iterations);
  int32_t* inl = ar.take<int32_t>(num_iterations);
  if (!ar.ok) {
    set_error("rdm_ransac_correspondences: workspace too small (%zu < %zu bytes)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  if (hyp_inliers) inl = hyp_inliers;
  if (n_corr < ransac_n) {  // Open3D returns an empty result (identity, fitness 0)
    RDM_HIP_CHECK(hipMemsetAsync(inl, 0, sizeof(int32_t) * num_iterations, st));
    RDM_HIP_CHECK(hipMemsetAsync(err2, 0, sizeof(double) * num_iterations, st));
  } else {
    RDM_REQUIRE(src_corr && ref_corr, "rdm_ransac_correspondences: null correspondences");
    hipLaunchKernelGGL(ransac_fit_kernel, dim3(ceil_div(num_iterations, 256)), dim3(256), 0, st, src_corr, ref_corr,
                       static_cast<int>(n_corr), ransac_n, num_iterations, static_cast<unsigned long long>(seed), hyp);
    hipLaunchKernelGGL(ransac_score_kernel, dim3(ceil_div(num_iterations, 4)), dim3(256), 0, st, src_corr, ref_corr,
                       static_cast<int>(n_corr), static_cast<double>(distance_threshold), num_iterations, hyp, inl, err2);
  }
  hipLaunchKernelGGL(ransac_best_kernel, dim3(1), dim3(1024), 0, st, num_iterations, hyp, inl, err2, transform, stats, inlier_rmse);
  return launch_status("rdm_ransac_correspondences");
}
