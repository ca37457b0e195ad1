// a14 -- log-domain Sinkhorn optimal transport with a learnable dustbin.
//
// Reference: geotransformer/modules/sinkhorn/learnable_sinkhorn.py:13-66.
// One workgroup per patch correspondence; the (valid rows + dustbin) x (valid cols + dustbin) block
// of the padded score matrix is staged through LDS once and then lives in REGISTERS for all
// iterations (each thread owns half a row and half a column).  Masked rows/columns are compacted
// away: in the reference they hold -1e12, so every exp() involving them underflows to exactly 0 and
// their own potentials evaluate to exactly 0 -- dropping them changes nothing but the summation
// order.  Masked entries of the output are written as fl(-1e12), the value the reference's
// fp32 arithmetic produces there.
// exp() inside the sums is the hardware exponential (__expf, arguments <= 0 after the max shift:
// relative error <= ~1e-6 on terms that matter); log() is the accurate one.
//   u = log_mu - logsumexp_j(Z + v)        logsumexp(x) = max + log(sum(exp(x - max)))
//   v = log_nu - logsumexp_i(Z + u)
//   out = ((Z + u) + v) - norm
#include <atomic>

#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "lockstep.h"

namespace {

using namespace rdm;

constexpr int kMaxSide = 128;

constexpr int kHalf = 66;   // each thread owns half a row and half a column: 65 entries, padded to an even count
constexpr int kPairs = kHalf / 2;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The iterations for a block whose compacted side fits 4 * PAIRS entries: thread t owns row (t>>1), columns
// [HALF*(t&1), HALF*(t&1) + HALF) with HALF = 2 * PAIRS, as float2 pairs in registers.  Three size classes (68, 100,
// 132) are instantiated: a patch with 60 valid points per side does half the work of a full one, and most patches
// are not full (the register arrays need static indexing, so the trip count cannot simply be a run-time bound).
// T threads share a line (row / column): T = 2 for the three large classes; small blocks -- most patches hold a few dozen
// real points -- use T = 4 (sides up to 64) and T = 8 (up to 32), so that all 256 threads work on the 64 / 32 lines there are
// instead of three quarters of them evaluating lines outside the block.
template <int PAIRS, int T = 2>
__device__ __forceinline__ void sinkhorn_iterate(const float* Z, int ldz, int nr, int nc, float norm, int iters, float* u, float* v,
                                                 const int* rows, const int* cols, int m, int n, float* O) {
  constexpr int HALF = 2 * PAIRS;
  static_assert(T == 2 || T == 4 || T == 8, "threads per line");
  const int tid = threadIdx.x, R = nr + 1, C = nc + 1;
  const int own = tid / T, half = tid % T, base = half * HALF;
  const float log_mu = own < nr ? norm : logf(static_cast<float>(nc)) + norm;
  const float log_nu = own < nc ? norm : logf(static_cast<float>(nr)) + norm;
  f32x2 zr[PAIRS], zc[PAIRS];  // -inf marks "outside the block": contributes exp(-inf) = 0
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const int c = base + i;
    zr[i >> 1][i & 1] = (own < R && c < C) ? Z[own * ldz + c] : -INFINITY;
    zc[i >> 1][i & 1] = (own < C && c < R) ? Z[c * ldz + own] : -INFINITY;
  }
  // log_m - logsumexp over this thread's half (z + pot) combined with the neighbouring lane's half
  auto update = [&](const f32x2 (&z)[PAIRS], const float* pot, float log_m) -> float {
    const f32x2* p2 = reinterpret_cast<const f32x2*>(pot + base);
    f32x2 t[PAIRS];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      t[i] = z[i] + p2[i];
      mx = fmaxf(mx, fmaxf(t[i].x, t[i].y));
    }
#pragma unroll
    for (int o = 1; o < T; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const f32x2 mx2 = {mx, mx}, l2e = {1.4426950408889634f, 1.4426950408889634f};
    f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const f32x2 d = (t[i] - mx2) * l2e;   // exp(x) = exp2(x log2 e): the hardware exponential (v_exp_f32), as __expf
      f32x2 e;
      e.x = __builtin_amdgcn_exp2f(d.x);
      e.y = __builtin_amdgcn_exp2f(d.y);
      sum2 += e;
    }
    float sum = sum2.x + sum2.y;
#pragma unroll
    for (int o = 1; o < T; o <<= 1) sum += __shfl_xor(sum, o, 64);
    return log_m - (mx + logf(sum));
  };

  // A side with all 128 points valid has 129 entries with the dustbin, one more than the 128 rows / columns the
  // thread pairs own.  Row / column 128 is then the dustbin line; wavefront 0 (row) and wavefront 1 (column)
  // evaluate it from LDS, three entries per lane, beside their own lines.
  const int wave = tid >> 6, lane = tid & 63;
  auto extra_line = [&](const float* zline, int zstride, const float* pot, int count, float log_m) -> float {
    float t[3], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = lane + 64 * i;
      t[i] = c < count ? zline[c * zstride] + pot[c] : -INFINITY;
      mx = fmaxf(mx, t[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) sum += __builtin_amdgcn_exp2f((t[i] - mx) * 1.4426950408889634f);
    sum = wave_sum(sum);
    return log_m - (mx + logf(sum));
  };
  const bool extra_row = R > 128, extra_col = C > 128;  // uniform over the workgroup
  const float log_mu_bin = logf(static_cast<float>(nc)) + norm, log_nu_bin = logf(static_cast<float>(nr)) + norm;

  for (int it = 0; it < iters; ++it) {
    const float un = update(zr, v, log_mu);  // u = log_mu - logsumexp_c(Z + v)
    if (half == 0 && own < R) u[own] = un;
    if (extra_row && wave == 0) {
      const float ue = extra_line(Z + 128 * ldz, 1, v, C, log_mu_bin);
      if (lane == 0) u[128] = ue;
    }
    __syncthreads();
    const float vn = update(zc, u, log_nu);  // v = log_nu - logsumexp_r(Z + u)
    if (half == 0 && own < C) v[own] = vn;
    if (extra_col && wave == 1) {
      const float ve = extra_line(Z + 128, ldz, u, R, log_nu_bin);
      if (lane == 0) v[128] = ve;
    }
    __syncthreads();
  }

  // dense output: fl(-1e12) everywhere, then the valid block
  const float masked = -1.0e12f;
  const int total = (m + 1) * (n + 1);
  for (int t = tid; t < total; t += 256) O[t] = masked;
  __syncthreads();
  if (own < R) {
    const float ur = u[own];
    const int64_t orow = static_cast<int64_t>(rows[own]) * (n + 1);
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
      const int c = base + i;
      if (c < C) O[orow + cols[c]] = ((zr[i >> 1][i & 1] + ur) + v[c]) - norm;
    }
  }
  if (extra_row) {  // the dustbin row of a full side (rows[128] = m)
    const float ur = u[128];
    const int64_t orow = static_cast<int64_t>(rows[128]) * (n + 1);
    for (int c = tid; c < C; c += 256) O[orow + cols[c]] = ((Z[128 * ldz + c] + ur) + v[c]) - norm;
  }
}

// 256 threads.  Thread t owns row (t>>1), columns [66*(t&1), 66*(t&1)+66) of the compacted score
// block in REGISTERS, and likewise half of column (t>>1): the 100 iterations touch LDS only for the
// broadcast potentials u, v.  A row's two halves are combined with one lane exchange.  The entries are held as
// float2 pairs: the adds, the shift by the maximum and the scaling by log2(e) are packed-fp32 instructions
// (v_pk_add_f32 / v_pk_mul_f32, two entries per issue slot) -- the loop is VALU-issue bound (per entry: add, max,
// subtract, multiply, v_exp_f32 at quarter rate, add), not LDS or latency bound.
__device__ __forceinline__ void sinkhorn_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* scores, int m, int n,
                                                       const unsigned char* row_mask,
                                                       const unsigned char* col_mask, const float* alpha_p,
                                                       int iters, float* out) {
  (void)blockIdx; (void)gridDim;
  extern __shared__ float lds[];
  __shared__ int rows[kMaxSide + 2], cols[kMaxSide + 2];
  __shared__ __attribute__((aligned(8))) float u[2 * kHalf], v[2 * kHalf];
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
  const int ldz = C | 1;
  float* Z = lds;
  const float alpha = *alpha_p;
  for (int t = tid; t < R * C; t += 256) {
    const int r = t / C, c = t % C;
    Z[r * ldz + c] = (r < nr && c < nc) ? S[static_cast<int64_t>(rows[r]) * n + cols[c]] : alpha;
  }
  const float norm = -logf(static_cast<float>(nr) + static_cast<float>(nc));
  for (int r = tid; r < 2 * kHalf; r += 256) {
    u[r] = 0.f;
    v[r] = 0.f;
  }
  __syncthreads();

  const int side = R > C ? R : C;
  if (side <= 32) sinkhorn_iterate<2, 8>(Z, ldz, nr, nc, norm, iters, u, v, rows, cols, m, n, O);
  else if (side <= 64) sinkhorn_iterate<8, 4>(Z, ldz, nr, nc, norm, iters, u, v, rows, cols, m, n, O);
  else if (side <= 68) sinkhorn_iterate<17>(Z, ldz, nr, nc, norm, iters, u, v, rows, cols, m, n, O);
  else if (side <= 100) sinkhorn_iterate<25>(Z, ldz, nr, nc, norm, iters, u, v, rows, cols, m, n, O);
  else sinkhorn_iterate<33>(Z, ldz, nr, nc, norm, iters, u, v, rows, cols, m, n, O);
}
__global__ __launch_bounds__(256) void sinkhorn_kernel(const float* scores, int m, int n,
                                                       const unsigned char* row_mask,
                                                       const unsigned char* col_mask, const float* alpha_p,
                                                       int iters, float* out) { sinkhorn_kernel_body(blockIdx, gridDim, scores, m, n, row_mask, col_mask, alpha_p, iters, out); }


}  // namespace

extern "C" int rdm_sinkhorn(const float* scores, int64_t batch, int64_t m, int64_t n, const uint8_t* row_mask,
                            const uint8_t* col_mask, const float* alpha, int iters, float* out, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(scores && row_mask && col_mask && alpha && out, "rdm_sinkhorn: null pointer");
  RDM_REQUIRE(batch >= 0 && m > 0 && n > 0 && m <= kMaxSide && n <= kMaxSide && iters >= 0,
              "rdm_sinkhorn: bad sizes (m=%lld n=%lld, max %d)", (long long)m, (long long)n, kMaxSide);
  if (batch == 0) return RDM_OK;
  const size_t lds = sizeof(float) * (static_cast<size_t>(m + 1) * ((n + 1) | 1) + 8);
  // the 129 x 129 fp32 tile (66.5 KB) needs more than the default 64 KB of dynamic LDS (set once per device)
  static std::atomic<uint64_t> attr_set{0};
  RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(sinkhorn_kernel), 160 * 1024 - 4096, attr_set));
  RDM_DUP_LOOP("sinkhorn")
  ::rdm::launch<sinkhorn_kernel_body, sinkhorn_kernel, 256>(dim3(static_cast<unsigned>(batch)), lds, static_cast<hipStream_t>(stream), scores, static_cast<int>(m), static_cast<int>(n), row_mask,
                     col_mask, alpha, iters, out);
  return launch_status("sinkhorn_kernel");
}
