// a14 -- log-domain Sinkhorn optimal transport with a learnable dustbin.
//
// Reference: geotransformer/modules/sinkhorn/learnable_sinkhorn.py:13-66.
// One workgroup per patch correspondence; the (valid rows + dustbin) x (valid cols + dustbin) block
// of the padded score matrix lives in LDS for all iterations.  Masked rows/columns are compacted
// away: in the reference they hold -1e12, so every exp() involving them underflows to exactly 0 and
// their own potentials evaluate to exactly 0 -- dropping them changes nothing but the summation
// order.  Masked entries of the output are written as fl(-1e12), the value the reference's
// fp32 arithmetic produces there.
//   u = log_mu - logsumexp_j(Z + v)        logsumexp(x) = max + log(sum(exp(x - max)))
//   v = log_nu - logsumexp_i(Z + u)
//   out = ((Z + u) + v) - norm
#include "../../include/rdmnet_hip.h"
#include "common.h"

namespace {

using namespace rdm;

constexpr int kMaxSide = 128;

__device__ __forceinline__ float group_max(float v, int gsize) {
  for (int o = gsize >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float group_sum(float v, int gsize) {
  for (int o = gsize >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void sinkhorn_kernel(const float* scores, int m, int n,
                                                       const unsigned char* row_mask,
                                                       const unsigned char* col_mask, const float* alpha_p,
                                                       int iters, float* out) {
  extern __shared__ float lds[];
  __shared__ int rows[kMaxSide + 1], cols[kMaxSide + 1];
  __shared__ int s_nr, s_nc;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* S = scores + static_cast<int64_t>(b) * m * n;
  const unsigned char* rm = row_mask + static_cast<int64_t>(b) * m;
  const unsigned char* cm = col_mask + static_cast<int64_t>(b) * n;
  float* O = out + static_cast<int64_t>(b) * (m + 1) * (n + 1);

  if (tid == 0) {
    int c = 0;
    for (int i = 0; i < m; ++i)
      if (rm[i]) rows[c++] = i;
    s_nr = c;
    rows[c] = m;  // dustbin row
  } else if (tid == 64) {
    int c = 0;
    for (int j = 0; j < n; ++j)
      if (cm[j]) cols[c++] = j;
    s_nc = c;
    cols[c] = n;  // dustbin column
  }
  __syncthreads();
  const int nr = s_nr, nc = s_nc, R = nr + 1, C = nc + 1;
  const int ldz = C | 1;  // odd stride: column walks are bank-conflict free
  float* Z = lds;
  float* u = Z + R * ldz;
  float* v = u + R;
  float* log_mu = v + C;
  float* log_nu = log_mu + R;
  const float alpha = *alpha_p;
  for (int t = tid; t < R * C; t += 256) {
    const int r = t / C, c = t % C;
    Z[r * ldz + c] = (r < nr && c < nc) ? S[static_cast<int64_t>(rows[r]) * n + cols[c]] : alpha;
  }
  const float norm = -logf(static_cast<float>(nr) + static_cast<float>(nc));
  for (int r = tid; r < R; r += 256) {
    log_mu[r] = r < nr ? norm : logf(static_cast<float>(nc)) + norm;
    u[r] = 0.f;
  }
  for (int c = tid; c < C; c += 256) {
    log_nu[c] = c < nc ? norm : logf(static_cast<float>(nr)) + norm;
    v[c] = 0.f;
  }
  __syncthreads();

  const int gc = C > 32 ? 64 : (C > 16 ? 32 : 16);  // lanes cooperating on one row
  const int gr = R > 32 ? 64 : (R > 16 ? 32 : 16);  // lanes cooperating on one column
  for (int it = 0; it < iters; ++it) {
    {
      const int grp = tid / gc, l = tid % gc, ngrp = 256 / gc;
      for (int r = grp; r < R; r += ngrp) {
        float mx = -INFINITY;
        for (int c = l; c < C; c += gc) mx = fmaxf(mx, Z[r * ldz + c] + v[c]);
        mx = group_max(mx, gc);
        float s = 0.f;
        for (int c = l; c < C; c += gc) s += expf(Z[r * ldz + c] + v[c] - mx);
        s = group_sum(s, gc);
        if (l == 0) u[r] = log_mu[r] - (mx + logf(s));
      }
    }
    __syncthreads();
    {
      const int grp = tid / gr, l = tid % gr, ngrp = 256 / gr;
      for (int c = grp; c < C; c += ngrp) {
        float mx = -INFINITY;
        for (int r = l; r < R; r += gr) mx = fmaxf(mx, Z[r * ldz + c] + u[r]);
        mx = group_max(mx, gr);
        float s = 0.f;
        for (int r = l; r < R; r += gr) s += expf(Z[r * ldz + c] + u[r] - mx);
        s = group_sum(s, gr);
        if (l == 0) v[c] = log_nu[c] - (mx + logf(s));
      }
    }
    __syncthreads();
  }

  // dense output: fl(-1e12) everywhere, then the valid block
  const float masked = -1.0e12f;
  const int total = (m + 1) * (n + 1);
  for (int t = tid; t < total; t += 256) O[t] = masked;
  __syncthreads();
  for (int t = tid; t < R * C; t += 256) {
    const int r = t / C, c = t % C;
    O[static_cast<int64_t>(rows[r]) * (n + 1) + cols[c]] = ((Z[r * ldz + c] + u[r]) + v[c]) - norm;
  }
}

}  // namespace

extern "C" int rdm_sinkhorn(const float* scores, int64_t batch, int64_t m, int64_t n, const uint8_t* row_mask,
                            const uint8_t* col_mask, const float* alpha, int iters, float* out, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(scores && row_mask && col_mask && alpha && out, "rdm_sinkhorn: null pointer");
  RDM_REQUIRE(batch >= 0 && m > 0 && n > 0 && m <= kMaxSide && n <= kMaxSide && iters >= 0,
              "rdm_sinkhorn: bad sizes (m=%lld n=%lld, max %d)", (long long)m, (long long)n, kMaxSide);
  if (batch == 0) return RDM_OK;
  const size_t lds = sizeof(float) * (static_cast<size_t>(m + 1) * ((n + 1) | 1) + 2 * (m + 1) + 2 * (n + 1) + 8);
  static bool attr_set = false;
  if (!attr_set) {  // the 129 x 129 fp32 tile (66.5 KB) needs more than the default 64 KB of dynamic LDS
    RDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
    attr_set = true;
  }
  hipLaunchKernelGGL(sinkhorn_kernel, dim3(static_cast<unsigned>(batch)), dim3(256), lds,
                     static_cast<hipStream_t>(stream), scores, static_cast<int>(m), static_cast<int>(n), row_mask,
                     col_mask, alpha, iters, out);
  return launch_status("sinkhorn_kernel");
}
