// a5 -- normalisation, activation and the neighbour pooling ops of the KPConv blocks.
//
//   GroupNorm over a stacked cloud (modules/kpconv/modules.py:33-50): the (N, C) feature matrix is
//     normalised as one (1, C, N) sample, i.e. statistics per group span ALL rows of BOTH clouds.
//     Three launches: per-block column partials (fp64, fixed order) -> per-channel scale/shift ->
//     fused apply (+ optional residual add, LeakyReLU(0.1), and the "row sum > 0" flag the next
//     KPConv needs, kpconv.py:113-114).
//   LayerNorm rows (transformer / vote MLP), with optional residual input and ReLU.
//   maxpool (modules/kpconv/functional.py:54-67) and nearest upsample + concat
//     (functional.py:6-22, experiments/backbone.py:131-143).
#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "internal.h"
#include "lockstep.h"

namespace {

using namespace rdm;

constexpr int kGnRowsPerBlock = 64;

// partial[blk][0][c] = sum, partial[blk][1][c] = sum of squares over this block's 64 rows.
// grid = (row blocks, column chunks of 256).  256 threads cover cw = min(C,256) columns x (256/cw)
// row lanes so that every wavefront reads whole contiguous row segments; the row lanes of a column
// are combined through LDS in fixed order (fp64 throughout).
__device__ __forceinline__ void gn_partial_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* x, int n, int c, int ld,
                                                         double* partial) {
  (void)blockIdx; (void)gridDim;
  __shared__ double red[2][256];
  const int r0 = blockIdx.x * kGnRowsPerBlock;
  const int r1 = min(n, r0 + kGnRowsPerBlock);
  const int cw = c < 256 ? c : 256;
  const int lanes = 256 / cw;
  const int col_in = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int col = blockIdx.y * 256 + col_in;
  double s = 0.0, ss = 0.0;
  if (rl < lanes && col < c)
    for (int r = r0 + rl; r < r1; r += lanes) {
      const double v = x[static_cast<int64_t>(r) * ld + col];
      s += v;
      ss += v * v;
    }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  if (threadIdx.x < cw && col < c) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < lanes; ++k) {
      a += red[0][k * cw + threadIdx.x];
      b += red[1][k * cw + threadIdx.x];
    }
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * c + col] = a;
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * c + col] = b;
  }
}
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* x, int n, int c, int ld,
                                                         double* partial) { gn_partial_kernel_body(blockIdx, gridDim, x, n, c, ld, partial); }


// One block per 64 columns (whole groups: C/groups divides 64): 16 lanes per column add the row-block
// partials in a fixed order, then scale[c] = rstd*gamma, shift[c] = beta - mean*rstd*gamma.
// COLS = 64: the layout above.  COLS = 16 (64 lanes per column) for the fine levels, whose GEMMs leave hundreds of partial
// rows for 32-128 columns: four times the workgroups and a quarter of the dependent load rounds per lane.  The per-column
// sums differ between the two only in the (fixed) order of fp64 additions.
// gn_scale_shift: the work of one such block (1024 threads, columns [col0, col0 + COLS)); the threads with lane == 0 return
// their column's pair, the others zeros.  Shared by the finalize kernel and the one-launch finalize + apply form below, which
// therefore produce the same scale / shift bits.
template <int COLS>
struct GnFinalizeLds {
  double sh[2][1024 / COLS][COLS];
};
template <int COLS>
__device__ __forceinline__ float2 gn_scale_shift(GnFinalizeLds<COLS>& L, int col0, const double* partial, int nblk, int n, int c,
                                                 int groups, const float* gamma, const float* beta, float eps) {
  constexpr int LANES = 1024 / COLS;
  const int ci = threadIdx.x % COLS, lane = threadIdx.x / COLS;
  const int col = col0 + ci;
  double s = 0.0, ss = 0.0;
  if (col < c)
    for (int b0 = lane; b0 < nblk; b0 += 4 * LANES) {  // four partial rows in flight, added in ascending order
      double p0[4], p1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + LANES * u;
        p0[u] = b < nblk ? partial[(static_cast<int64_t>(b) * 2 + 0) * c + col] : 0.0;
        p1[u] = b < nblk ? partial[(static_cast<int64_t>(b) * 2 + 1) * c + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s += p0[u];
        ss += p1[u];
      }
    }
  L.sh[0][lane][ci] = s;
  L.sh[1][lane][ci] = ss;
  __syncthreads();
  if (lane == 0) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < LANES; ++k) {
      a += L.sh[0][k][ci];
      b += L.sh[1][k][ci];
    }
    L.sh[0][0][ci] = a;
    L.sh[1][0][ci] = b;
  }
  __syncthreads();
  float2 out = make_float2(0.f, 0.f);
  if (lane == 0 && col < c) {
    const int cpg = c / groups;
    const int g0 = (ci / cpg) * cpg;
    double gs = 0.0, gss = 0.0;
    for (int k = 0; k < cpg; ++k) {
      gs += L.sh[0][0][g0 + k];
      gss += L.sh[1][0][g0 + k];
    }
    const double cnt = static_cast<double>(n) * cpg;
    const double mean = gs / cnt;
    double var = gss / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + static_cast<double>(eps));
    const double gsc = rstd * static_cast<double>(gamma[col]);
    out.x = static_cast<float>(gsc);
    out.y = static_cast<float>(static_cast<double>(beta[col]) - mean * gsc);
  }
  return out;
}
template <int COLS>
__device__ __forceinline__ void gn_finalize_kernel_body(const dim3 blockIdx, const dim3 gridDim, const double* partial, int nblk, int n,
                                                            int c, int groups, const float* gamma,
                                                            const float* beta, float eps, float* scale,
                                                            float* shift) {
  (void)blockIdx; (void)gridDim;
  __shared__ GnFinalizeLds<COLS> L;
  const float2 v = gn_scale_shift<COLS>(L, blockIdx.x * COLS, partial, nblk, n, c, groups, gamma, beta, eps);
  const int col = blockIdx.x * COLS + threadIdx.x % COLS;
  if (threadIdx.x / COLS == 0 && col < c) {
    scale[col] = v.x;
    shift[col] = v.y;
  }
}
template <int COLS>
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const double* partial, int nblk, int n,
                                                            int c, int groups, const float* gamma,
                                                            const float* beta, float eps, float* scale,
                                                            float* shift) { gn_finalize_kernel_body<COLS>(blockIdx, gridDim, partial, nblk, n, c, groups, gamma, beta, eps, scale, shift); }


// Finalize + apply in ONE launch for the coarse levels (a few thousand rows at most): workgroup (slab, chunk) computes the scale /
// shift of its 64 columns exactly as gn_finalize_kernel<64> does (a few dozen partial rows: cheap enough to repeat per row chunk)
// and applies them to 256 rows of that slab with gn_apply_wide_kernel's arithmetic -- same bits as the two launches, one
// dependent launch (~4.5 us on a pair's critical path) less per GroupNorm.  No positive-row flag (a row spans several slabs).
constexpr int kGnFusedRows = 256;
__device__ __forceinline__ void gn_finalize_apply_kernel_body(const dim3 blockIdx, const dim3 gridDim, const double* partial, int nblk, const float* x, int n, int c, int ldx,
                                                                  int groups, const float* gamma, const float* beta, float eps,
                                                                  const float* res, int ldr, int act, float* y, int ldy) {
  (void)blockIdx; (void)gridDim;
  __shared__ GnFinalizeLds<64> L;
  __shared__ float sc_l[64], sh_l[64];
  const int col0 = blockIdx.x * 64;
  const float2 v = gn_scale_shift<64>(L, col0, partial, nblk, n, c, groups, gamma, beta, eps);
  if (threadIdx.x < 64) {  // (lane == 0: thread index = column within the slab)
    sc_l[threadIdx.x] = v.x;
    sh_l[threadIdx.x] = v.y;
  }
  __syncthreads();
  const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;  // 16 float4 per slab row, 64 rows per pass
  const int col = col0 + 4 * q;
  if (col >= c) return;
  const float4 sc = *reinterpret_cast<const float4*>(sc_l + 4 * q), sh = *reinterpret_cast<const float4*>(sh_l + 4 * q);
  const int r1 = min(n, (static_cast<int>(blockIdx.y) + 1) * kGnFusedRows);
  for (int row = blockIdx.y * kGnFusedRows + rl; row < r1; row += 64) {
    const float4 xv = *reinterpret_cast<const float4*>(x + static_cast<int64_t>(row) * ldx + col);
    float o[4] = {__builtin_fmaf(xv.x, sc.x, sh.x), __builtin_fmaf(xv.y, sc.y, sh.y), __builtin_fmaf(xv.z, sc.z, sh.z),
                  __builtin_fmaf(xv.w, sc.w, sh.w)};
    if (res) {
      const float4 rv = *reinterpret_cast<const float4*>(res + static_cast<int64_t>(row) * ldr + col);
      o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (act == 2) o[k] = o[k] > 0.f ? o[k] : 0.1f * o[k];
      else if (act == 1) o[k] = o[k] > 0.f ? o[k] : 0.f;
    }
    *reinterpret_cast<float4*>(y + static_cast<int64_t>(row) * ldy + col) = make_float4(o[0], o[1], o[2], o[3]);
  }
}
__global__ __launch_bounds__(1024) void gn_finalize_apply_kernel(const double* partial, int nblk, const float* x, int n, int c, int ldx,
                                                                  int groups, const float* gamma, const float* beta, float eps,
                                                                  const float* res, int ldr, int act, float* y, int ldy) { gn_finalize_apply_kernel_body(blockIdx, gridDim, partial, nblk, x, n, c, ldx, groups, gamma, beta, eps, res, ldr, act, y, ldy); }


// y = act(x*scale + shift (+ res)); optional positive-row flag.  One wavefront per row.
__device__ __forceinline__ void gn_apply_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* x, int n, int c, int ldx,
                                                       const float* scale, const float* shift,
                                                       const float* res, int ldr, int act, float* y,
                                                       int ldy, unsigned char* positive) {
  (void)blockIdx; (void)gridDim;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  float rs = 0.f;
  for (int col = lane; col < c; col += 64) {
    float v = x[static_cast<int64_t>(row) * ldx + col] * scale[col] + shift[col];
    if (res) v += res[static_cast<int64_t>(row) * ldr + col];
    if (act == 2) v = v > 0.f ? v : 0.1f * v;
    else if (act == 1) v = v > 0.f ? v : 0.f;
    y[static_cast<int64_t>(row) * ldy + col] = v;
    rs += v;
  }
  if (positive) {
    rs = wave_sum(rs);
    if (lane == 0) positive[row] = rs > 0.f ? 1 : 0;
  }
}
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* x, int n, int c, int ldx,
                                                       const float* scale, const float* shift,
                                                       const float* res, int ldr, int act, float* y,
                                                       int ldy, unsigned char* positive) { gn_apply_kernel_body(blockIdx, gridDim, x, n, c, ldx, scale, shift, res, ldr, act, y, ldy, positive); }


// Same without the row flag, for wide matrices with few rows (coarse levels: 800 x 2048): one thread per
// float4, grid over (rows, column quads) so that the launch has enough wavefronts to cover the latency.
// `positive` (optional, needs c4 a power of two <= 64 so that a row lies inside one wavefront): the row-sum flag of
// gn_apply_kernel with the SAME summation tree (column index bits from high to low: gn_apply_kernel's lane-strided partial
// sums and xor-butterfly), so both kernels set the same flags bit for bit.
__device__ __forceinline__ void gn_apply_wide_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* x, int n, int c4, int ldx, const float* scale,
                                                            const float* shift, const float* res, int ldr, int act,
                                                            float* y, int ldy, unsigned char* positive) {
  (void)blockIdx; (void)gridDim;
  const int64_t t = blockIdx.x * 256ll + threadIdx.x;
  const bool live = t < static_cast<int64_t>(n) * c4;
  if (!live && !positive) return;
  float rs4[4] = {0.f, 0.f, 0.f, 0.f};
  const int row = live ? static_cast<int>(t / c4) : 0, col = live ? static_cast<int>(t % c4) * 4 : 0;
  if (live) {
  const float4 xv = *reinterpret_cast<const float4*>(x + static_cast<int64_t>(row) * ldx + col);
  const float4 sc = *reinterpret_cast<const float4*>(scale + col), sh = *reinterpret_cast<const float4*>(shift + col);
  float v[4] = {__builtin_fmaf(xv.x, sc.x, sh.x), __builtin_fmaf(xv.y, sc.y, sh.y), __builtin_fmaf(xv.z, sc.z, sh.z),
                __builtin_fmaf(xv.w, sc.w, sh.w)};  // (what the compiler contracted x * scale + shift to since round 1)
  if (res) {
    const float4 rv = *reinterpret_cast<const float4*>(res + static_cast<int64_t>(row) * ldr + col);
    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (act == 2) v[k] = v[k] > 0.f ? v[k] : 0.1f * v[k];
    else if (act == 1) v[k] = v[k] > 0.f ? v[k] : 0.f;
  }
  *reinterpret_cast<float4*>(y + static_cast<int64_t>(row) * ldy + col) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k) rs4[k] = v[k];
  }
  if (positive) {  // (every lane of the wavefront takes part in the exchanges; rows never straddle wavefronts)
    for (int o = c4 >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) rs4[k] += __shfl_xor(rs4[k], o, 64);
    const float rs = (rs4[0] + rs4[2]) + (rs4[1] + rs4[3]);
    if (live && col == 0) positive[row] = rs > 0.f ? 1 : 0;
  }
}
__global__ __launch_bounds__(256) void gn_apply_wide_kernel(const float* x, int n, int c4, int ldx, const float* scale,
                                                            const float* shift, const float* res, int ldr, int act,
                                                            float* y, int ldy, unsigned char* positive) { gn_apply_wide_kernel_body(blockIdx, gridDim, x, n, c4, ldx, scale, shift, res, ldr, act, y, ldy, positive); }


// y = act(LayerNorm(x (+ res)) * gamma + beta); one wavefront per row, c <= 2048
__device__ __forceinline__ void layernorm_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* x, int n, int c, int ldx,
                                                        const float* res, int ldr,
                                                        const float* gamma, const float* beta,
                                                        float eps, int act, float* y, int ldy) {
  (void)blockIdx; (void)gridDim;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  float v[32];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int col = lane + 64 * i;
    v[i] = 0.f;
    if (col < c) {
      v[i] = x[static_cast<int64_t>(row) * ldx + col];
      if (res) v[i] += res[static_cast<int64_t>(row) * ldr + col];
      s += v[i];
    }
  }
  const float mean = wave_sum(s) / static_cast<float>(c);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int col = lane + 64 * i;
    if (col < c) {
      const float d = v[i] - mean;
      ss += d * d;
    }
  }
  const float rstd = 1.0f / __fsqrt_rn(wave_sum(ss) / static_cast<float>(c) + eps);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int col = lane + 64 * i;
    if (col < c) {
      float o = (v[i] - mean) * rstd * gamma[col] + beta[col];
      if (act == 1) o = o > 0.f ? o : 0.f;
      y[static_cast<int64_t>(row) * ldy + col] = o;
    }
  }
}
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, int n, int c, int ldx,
                                                        const float* res, int ldr,
                                                        const float* gamma, const float* beta,
                                                        float eps, int act, float* y, int ldy) { layernorm_kernel_body(blockIdx, gridDim, x, n, c, ldx, res, ldr, gamma, beta, eps, act, y, ldy); }


// out[m, c] = max_h x[idx[m,h], c], pad rows count as zeros.  One wavefront per (row, 256 channels).
// `order` (optional): cell-sorted query records of the level's search grid (query row = int bits of .w).  Queries are
// then visited in cell order and workgroups re-mapped so that each XCD (workgroup b runs on XCD b % 8, own 4 MB L2) owns a
// contiguous range of them -- a slab of space, whose support rows (one eighth of the tensor plus a halo) stay in that L2;
// in row order (hash-map order, spatially random) every XCD streams the whole tensor from the fabric.
// LPR = lanes per feature row (a lane holds 4 channels): rows of 64 / 128 channels take 16 / 32 lanes, so a wavefront serves
// four / two queries at once (round 4: with one query per wavefront three quarters of the lanes idled on the 64-channel pool of
// the first strided block, and a wavefront walked its 65 slots in nine dependent trips of eight rows: 47 us); 256 channels and
// more: one query per wavefront and 256-channel chunk (blockIdx.y).  Sixteen slots per trip, and a shadow slot costs no load:
// it contributes the zero row, i.e. max(., 0) once at the end.
// SPLIT (coarse levels: a few hundred queries of 256+ channels leave most CUs idle, and a wavefront's five dependent trips set
// the kernel's time): the four wavefronts of a workgroup share ONE (query, 256-channel chunk), each takes every fourth
// group of slots, and the four partial maxima meet in LDS (max is exact: the result does not depend on the split).
template <int LPR, bool SPLIT = false>
__device__ __forceinline__ void gather_max_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* x, int ns, int c, int ldx,
                                                         const int64_t* idx, int m_total, int h, int ldi,
                                                         const int32_t* width, float* y, int ldy, const float4* order, int i32) {
  (void)blockIdx; (void)gridDim;
  constexpr int QPW = 64 / LPR, U = SPLIT ? 8 : 16;
  static_assert(!SPLIT || LPR == 64, "the slot split is for one query per wavefront");
  __shared__ float4 part[SPLIT ? 3 * 64 : 1];
  int blk = blockIdx.x;
  if (order && gridDim.x >= 16) {  // bijective: XCD x takes logical workgroups [start_x, start_x + count_x)
    const int nblk = gridDim.x, q = nblk / 8, rr = nblk % 8, xcd = blk % 8, within = blk / 8;
    blk = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + within;
  }
  const int lane = threadIdx.x & 63, sub = lane / LPR, l = lane % LPR, wave = threadIdx.x >> 6;
  const int unit = SPLIT ? blk : (blk * 4 + wave) * QPW + sub;
  const bool active = unit < m_total;
  const int m = active ? (order ? __float_as_int(order[unit].w) : unit) : 0;
  const int c0 = blockIdx.y * 256 + l * 4;
  int H = h;
  if (width) H = min(H, *width);
  if (!active || c0 >= c) H = 0;  // (idle lanes run the loop without loads; the trip count below is wavefront-uniform)
  float4 best = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
  bool shadow = false;
  const long long row_idx = static_cast<long long>(m) * ldi;  // element offset (int64 or int32 elements)
  const int Hmax = width ? min(h, *width) : h;
  for (int k0 = SPLIT ? wave * U : 0; k0 < Hmax; k0 += SPLIT ? 4 * U : U) {  // U neighbour rows in flight (index -> row is a dependent load pair)
    int id[U];  // support row, or -1
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t raw = k0 + u < H ? ld_index(idx, row_idx + k0 + u, i32) : -1;
      const bool real = raw >= 0 && raw < ns;
      shadow = shadow || (k0 + u < H && !real);
      id[u] = real ? static_cast<int>(raw) : -1;
    }
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float4 t = *reinterpret_cast<const float4*>(x + static_cast<int64_t>(max(id[u], 0)) * ldx + (c0 < c ? c0 : 0));
      v[u] = make_float4(id[u] >= 0 ? t.x : best.x, id[u] >= 0 ? t.y : best.y, id[u] >= 0 ? t.z : best.z, id[u] >= 0 ? t.w : best.w);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      best.x = fmaxf(best.x, v[u].x);
      best.y = fmaxf(best.y, v[u].y);
      best.z = fmaxf(best.z, v[u].z);
      best.w = fmaxf(best.w, v[u].w);
    }
  }
  if constexpr (SPLIT) {
    if (wave > 0) part[(wave - 1) * 64 + lane] = make_float4(shadow ? fmaxf(best.x, 0.f) : best.x, shadow ? fmaxf(best.y, 0.f) : best.y,
                                                            shadow ? fmaxf(best.z, 0.f) : best.z, shadow ? fmaxf(best.w, 0.f) : best.w);
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const float4 t = part[w * 64 + lane];
      best.x = fmaxf(best.x, t.x); best.y = fmaxf(best.y, t.y); best.z = fmaxf(best.z, t.z); best.w = fmaxf(best.w, t.w);
    }
  }
  if (shadow) {  // a shadow neighbour is the zero row (functional.py:54-67)
    best.x = fmaxf(best.x, 0.f);
    best.y = fmaxf(best.y, 0.f);
    best.z = fmaxf(best.z, 0.f);
    best.w = fmaxf(best.w, 0.f);
  }
  if (active && c0 < c) *reinterpret_cast<float4*>(y + static_cast<int64_t>(m) * ldy + c0) = best;
}
template <int LPR, bool SPLIT = false>
__global__ __launch_bounds__(256) void gather_max_kernel(const float* x, int ns, int c, int ldx,
                                                         const int64_t* idx, int m_total, int h, int ldi,
                                                         const int32_t* width, float* y, int ldy, const float4* order, int i32) { gather_max_kernel_body<LPR, SPLIT>(blockIdx, gridDim, x, ns, c, ldx, idx, m_total, h, ldi, width, y, ldy, order, i32); }


// y[m, 0:c1] = coarse[idx[m,0]] (pad -> 0), y[m, c1:c1+c2] = skip[m], y[m, c1+c2:ldy] = 0
__device__ __forceinline__ void upsample_concat_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* coarse, int n_coarse, int c1, int ld1,
                                       const int64_t* idx, int ldi, const float* skip, int c2, int ld2,
                                       int m_total, float* y, int ldy) {
  (void)blockIdx; (void)gridDim;
  const int m = blockIdx.x;
  const int64_t id = idx[static_cast<int64_t>(m) * ldi];
  const bool ok = id >= 0 && id < n_coarse;
  for (int col = threadIdx.x; col < ldy; col += blockDim.x) {
    float v = 0.f;
    if (col < c1) v = ok ? coarse[id * ld1 + col] : 0.f;
    else if (col < c1 + c2) v = skip[static_cast<int64_t>(m) * ld2 + (col - c1)];
    y[static_cast<int64_t>(m) * ldy + col] = v;
  }
}
__global__ void upsample_concat_kernel(const float* coarse, int n_coarse, int c1, int ld1,
                                       const int64_t* idx, int ldi, const float* skip, int c2, int ld2,
                                       int m_total, float* y, int ldy) { upsample_concat_kernel_body(blockIdx, gridDim, coarse, n_coarse, c1, ld1, idx, ldi, skip, c2, ld2, m_total, y, ldy); }


// y[i, :] = x[idx[i], :] as raw 32-bit words; rows with idx outside [0, n_src) become zeros
__device__ __forceinline__ void gather_rows_kernel_body(const dim3 blockIdx, const dim3 gridDim, const uint32_t* x, int n_src, int c, int ldx, const int64_t* idx, int m,
                                   uint32_t* y, int ldy) {
  (void)blockIdx; (void)gridDim;
  const int row = blockIdx.x;
  const int64_t id = idx[row];
  const bool ok = id >= 0 && id < n_src;
  for (int col = threadIdx.x; col < c; col += blockDim.x)
    y[static_cast<int64_t>(row) * ldy + col] = ok ? x[id * ldx + col] : 0u;
}
__global__ void gather_rows_kernel(const uint32_t* x, int n_src, int c, int ldx, const int64_t* idx, int m,
                                   uint32_t* y, int ldy) { gather_rows_kernel_body(blockIdx, gridDim, x, n_src, c, ldx, idx, m, y, ldy); }


// Up to four independent row gathers in one launch: a 1-D grid, workgroup -> (gather, row) through first_row.
struct GatherItem {
  const uint32_t* x;
  int n_src, c, ldx;
  const int64_t* idx;
  uint32_t* y;
  int ldy;
  int m, lanes;  // rows; lanes per row: the power of two >= c (at most the workgroup) -- a workgroup copies threads / lanes rows
};
struct GatherBatch {
  GatherItem item[4];
  int first_block[5];
  int n;
};
// (Round 5: several rows per workgroup.  The patch-point gather of a pair is 2 x 16 384 rows of THREE words -- one workgroup per
// row was 32 768 workgroups for 400 KB, 70 us a pair; rows of 3 words now go 16 to a wavefront.)
__device__ __forceinline__ void gather_rows_multi_kernel_body(const dim3 blockIdx, const dim3 gridDim, GatherBatch b) {
  (void)blockIdx; (void)gridDim;
  int it = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) it += (k < b.n && static_cast<int>(blockIdx.x) >= b.first_block[k]) ? 1 : 0;
  const GatherItem& g = b.item[it];
  const int lanes = g.lanes, per_block = blockDim.x / lanes;
  const int row = (blockIdx.x - b.first_block[it]) * per_block + threadIdx.x / lanes;
  if (row >= g.m) return;
  const int64_t id = g.idx[row];
  const bool ok = id >= 0 && id < g.n_src;
  for (int col = threadIdx.x % lanes; col < g.c; col += lanes)
    g.y[static_cast<int64_t>(row) * g.ldy + col] = ok ? g.x[id * g.ldx + col] : 0u;
}
__global__ void gather_rows_multi_kernel(GatherBatch b) { gather_rows_multi_kernel_body(blockIdx, gridDim, b); }


}  // namespace

int rdm::gather_rows_multi(int n, const void* const* x, const int64_t* n_src, const int64_t* words, const int64_t* ldx,
                           const int64_t* const* idx, const int64_t* m, void* const* y, const int64_t* ldy, void* stream) {
  RDM_REQUIRE(n >= 1 && n <= 4, "gather_rows_multi: 1..4 gathers");
  GatherBatch b;
  b.n = n;
  b.first_block[0] = 0;
  int64_t max_words = 0;
  for (int k = 0; k < n; ++k) max_words = std::max(max_words, words[k]);
  const int threads = max_words >= 256 ? 256 : (max_words >= 128 ? 128 : 64);
  for (int k = 0; k < n; ++k) {
    RDM_REQUIRE(x[k] && idx[k] && y[k] && words[k] > 0 && m[k] >= 0, "gather_rows_multi: bad arguments");
    int lanes = 1;
    while (lanes < words[k] && lanes < threads) lanes <<= 1;
    b.item[k] = GatherItem{static_cast<const uint32_t*>(x[k]), static_cast<int>(n_src[k]), static_cast<int>(words[k]),
                           static_cast<int>(ldx[k]), idx[k], static_cast<uint32_t*>(y[k]), static_cast<int>(ldy[k]),
                           static_cast<int>(m[k]), lanes};
    b.first_block[k + 1] = b.first_block[k] + static_cast<int>(ceil_div<int64_t>(m[k], threads / lanes));
  }
  if (b.first_block[n] == 0) return RDM_OK;
  const dim3 grid(static_cast<unsigned>(b.first_block[n]));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (threads == 256) ::rdm::launch<gather_rows_multi_kernel_body, gather_rows_multi_kernel, 256>(grid, 0, st, b);
  else if (threads == 128) ::rdm::launch<gather_rows_multi_kernel_body, gather_rows_multi_kernel, 128>(grid, 0, st, b);
  else ::rdm::launch<gather_rows_multi_kernel_body, gather_rows_multi_kernel, 64>(grid, 0, st, b);
  return launch_status("gather_rows_multi_kernel");
}

extern "C" int rdm_gather_rows(const void* x, int64_t n_src, int64_t words, int64_t ldx, const int64_t* idx,
                               int64_t m, void* y, int64_t ldy, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(x && idx && y && words > 0, "rdm_gather_rows: bad arguments");
  if (m == 0) return RDM_OK;
  const int threads = words >= 256 ? 256 : (words >= 128 ? 128 : 64);
  if (threads == 256) ::rdm::launch<gather_rows_kernel_body, gather_rows_kernel, 256>(dim3(static_cast<unsigned>(m)), 0, static_cast<hipStream_t>(stream), static_cast<const uint32_t*>(x), static_cast<int>(n_src),
                     static_cast<int>(words), static_cast<int>(ldx), idx, static_cast<int>(m),
                     static_cast<uint32_t*>(y), static_cast<int>(ldy));
  else if (threads == 128) ::rdm::launch<gather_rows_kernel_body, gather_rows_kernel, 128>(dim3(static_cast<unsigned>(m)), 0, static_cast<hipStream_t>(stream), static_cast<const uint32_t*>(x), static_cast<int>(n_src),
                     static_cast<int>(words), static_cast<int>(ldx), idx, static_cast<int>(m),
                     static_cast<uint32_t*>(y), static_cast<int>(ldy));
  else ::rdm::launch<gather_rows_kernel_body, gather_rows_kernel, 64>(dim3(static_cast<unsigned>(m)), 0, static_cast<hipStream_t>(stream), static_cast<const uint32_t*>(x), static_cast<int>(n_src),
                     static_cast<int>(words), static_cast<int>(ldx), idx, static_cast<int>(m),
                     static_cast<uint32_t*>(y), static_cast<int>(ldy));
  return launch_status("gather_rows_kernel");
}

extern "C" size_t rdm_group_norm_workspace_bytes(int64_t n, int64_t c) {
  const size_t nblk = static_cast<size_t>(rdm::ceil_div<int64_t>(n > 0 ? n : 1, kGnRowsPerBlock));
  return rdm::align_up(nblk * 2 * c * sizeof(double)) + rdm::align_up(2 * c * sizeof(float));
}

int rdm::group_norm_finish(const double* partial_in, int nblk, const float* x, int64_t n, int64_t c, int64_t ldx,
                           int groups, const float* gamma, const float* beta, float eps, const float* residual,
                           int64_t ldr, int act, float* y, int64_t ldy, uint8_t* positive, void* ws, size_t ws_bytes,
                           void* stream, int form) {
  Arena ar(ws, ws_bytes);
  const int own_blk = static_cast<int>(ceil_div<int64_t>(n, kGnRowsPerBlock));
  double* partial = ar.take<double>(static_cast<size_t>(own_blk) * 2 * c);
  float* ss = ar.take<float>(2 * c);
  if (!ar.ok) {
    set_error("rdm_group_norm: workspace too small (%zu < %zu)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double* use = partial_in;
  if (nblk <= 0) {
    ::rdm::launch<gn_partial_kernel_body, gn_partial_kernel, 256>(dim3(own_blk, ceil_div<int64_t>(c, 256)), 0, st, x,
                       static_cast<int>(n), static_cast<int>(c), static_cast<int>(ldx), partial);
    use = partial;
    nblk = own_blk;
  }
  const bool vec_ok = c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (!residual || ldr % 4 == 0) &&
                      (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                      (!residual || (reinterpret_cast<uintptr_t>(residual) & 15) == 0);
  static const bool fin64 = ::rdm::dev_knob("RDM_GN_FINALIZE_64") != nullptr;  // developer knob (A/B): always 64 columns per workgroup
  const bool narrow = !fin64 && nblk >= 128 && c / groups <= 16 && 16 % (c / groups) == 0;
  // coarse levels: finalize + apply as one launch (see gn_finalize_apply_kernel); form 1 = always the separate launches
  if (form != 1 && !narrow && !positive && vec_ok && c % 64 == 0 && n <= 4096) {
    RDM_DUP_LOOP("gnfin")
    ::rdm::launch<gn_finalize_apply_kernel_body, gn_finalize_apply_kernel, 1024>(dim3(static_cast<unsigned>(c / 64), static_cast<unsigned>(ceil_div<int64_t>(n, kGnFusedRows))), 0, st, use, nblk, x, static_cast<int>(n), static_cast<int>(c), static_cast<int>(ldx), groups, gamma, beta,
                       eps, residual, static_cast<int>(ldr), act, y, static_cast<int>(ldy));
    return launch_status("gn_finalize_apply_kernel");
  }
  {
    RDM_DUP_LOOP("gnfin")
    if (narrow)
      ::rdm::launch<gn_finalize_kernel_body<16>, gn_finalize_kernel<16>, 1024>(dim3(ceil_div<int64_t>(c, 16)), 0, st, use, nblk,
                         static_cast<int>(n), static_cast<int>(c), groups, gamma, beta, eps, ss, ss + c);
    else
    ::rdm::launch<gn_finalize_kernel_body<64>, gn_finalize_kernel<64>, 1024>(dim3(ceil_div<int64_t>(c, 64)), 0, st, use, nblk,
                       static_cast<int>(n), static_cast<int>(c), groups, gamma, beta, eps, ss, ss + c);
  }
  // 16-byte accesses whenever the layout allows; with the positive-row flag only where a row lies inside one wavefront AND
  // gn_apply_kernel's summation tree can be reproduced (up to 128 columns: at most two values per lane there)
  static const bool narrow_rows = ::rdm::dev_knob("RDM_GN_APPLY_ROWS") != nullptr;  // developer knob (A/B): one wavefront per row below 256 columns
  const int64_t c4 = c / 4;
  const bool flag_ok = c4 == 8 || c4 == 16 || c4 == 32;
  const bool wide = vec_ok && (narrow_rows ? (!positive && c >= 256) : (!positive || flag_ok));
  RDM_DUP_LOOP("gnapply")
  if (wide)
    ::rdm::launch<gn_apply_wide_kernel_body, gn_apply_wide_kernel, 256>(dim3(ceil_div<int64_t>(n * (c / 4), 256)), 0, st, x, static_cast<int>(n),
                       static_cast<int>(c / 4), static_cast<int>(ldx), ss, ss + c, residual, static_cast<int>(ldr), act, y,
                       static_cast<int>(ldy), positive);
  else
    ::rdm::launch<gn_apply_kernel_body, gn_apply_kernel, 256>(dim3(ceil_div<int64_t>(n, 4)), 0, st, x,
                       static_cast<int>(n), static_cast<int>(c), static_cast<int>(ldx), ss, ss + c, residual,
                       static_cast<int>(ldr), act, y, static_cast<int>(ldy), positive);
  return launch_status("group_norm kernels");
}

extern "C" int rdm_group_norm(const float* x, int64_t n, int64_t c, int64_t ldx, int groups,
                              const float* gamma, const float* beta, float eps, const float* residual,
                              int64_t ldr, int act, float* y, int64_t ldy, uint8_t* positive, void* ws,
                              size_t ws_bytes, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(x && gamma && beta && y, "rdm_group_norm: null pointer");
  RDM_REQUIRE(n >= 0 && c > 0 && groups > 0 && c % groups == 0 && c <= 4096 && 64 % (c / groups) == 0,
              "rdm_group_norm: bad sizes (n=%lld c=%lld groups=%d)", (long long)n, (long long)c, groups);
  if (n == 0) return RDM_OK;
  return group_norm_finish(nullptr, 0, x, n, c, ldx, groups, gamma, beta, eps, residual, ldr, act, y, ldy, positive, ws,
                           ws_bytes, stream);
}

extern "C" int rdm_group_norm_form(const float* x, int64_t n, int64_t c, int64_t ldx, int groups, const float* gamma,
                                   const float* beta, float eps, const float* residual, int64_t ldr, int act, float* y, int64_t ldy,
                                   uint8_t* positive, void* ws, size_t ws_bytes, int form, void* stream) {
  RDM_REQUIRE(form >= 0 && form <= 1, "rdm_group_norm_form: form must be 0 (the library's choice) or 1 (statistics, finalize, apply as three launches)");
  using namespace rdm;
  RDM_REQUIRE(x && gamma && beta && y, "rdm_group_norm: null pointer");
  RDM_REQUIRE(n >= 0 && c > 0 && groups > 0 && c % groups == 0 && c <= 4096 && 64 % (c / groups) == 0,
              "rdm_group_norm: bad sizes (n=%lld c=%lld groups=%d)", (long long)n, (long long)c, groups);
  if (n == 0) return RDM_OK;
  return group_norm_finish(nullptr, 0, x, n, c, ldx, groups, gamma, beta, eps, residual, ldr, act, y, ldy, positive, ws, ws_bytes,
                           stream, form);
}

extern "C" int rdm_layer_norm(const float* x, int64_t n, int64_t c, int64_t ldx, const float* residual,
                              int64_t ldr, const float* gamma, const float* beta, float eps, int act,
                              float* y, int64_t ldy, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(x && gamma && beta && y, "rdm_layer_norm: null pointer");
  RDM_REQUIRE(n >= 0 && c > 0 && c <= 2048, "rdm_layer_norm: bad sizes");
  if (n == 0) return RDM_OK;
  RDM_DUP_LOOP("ln")
  ::rdm::launch<layernorm_kernel_body, layernorm_kernel, 256>(dim3(ceil_div<int64_t>(n, 4)), 0, static_cast<hipStream_t>(stream), x, static_cast<int>(n), static_cast<int>(c),
                     static_cast<int>(ldx), residual, static_cast<int>(ldr), gamma, beta, eps, act, y,
                     static_cast<int>(ldy));
  return launch_status("layernorm_kernel");
}

int rdm::gather_max_ordered(const float* x, int64_t n_s, int64_t c, int64_t ldx, const int64_t* idx, int64_t m, int64_t h,
                            int64_t ldi, const int32_t* width, float* y, int64_t ldy, const float* order_records, int i32, void* stream) {
  RDM_REQUIRE(x && idx && y, "rdm_gather_max: null pointer");
  RDM_REQUIRE(c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && h > 0, "rdm_gather_max: bad sizes");
  if (m == 0) return RDM_OK;
  static const bool no_order = ::rdm::dev_knob("RDM_NO_POOL_ORDER") != nullptr;  // developer knob (A/B): row order
  if (no_order) order_records = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float4* ord = reinterpret_cast<const float4*>(order_records);
  const int ns = static_cast<int>(n_s), ci = static_cast<int>(c), lx = static_cast<int>(ldx), mi = static_cast<int>(m), hi = static_cast<int>(h),
            li = static_cast<int>(ldi), ly = static_cast<int>(ldy);
  RDM_DUP_LOOP("pool") {
    if (c <= 64)  // four queries per wavefront
      ::rdm::launch<gather_max_kernel_body<16>, gather_max_kernel<16>, 256>(dim3(ceil_div<int64_t>(m, 16), 1), 0, st, x, ns, ci, lx, idx, mi, hi, li, width, y, ly, ord, i32);
    else if (c <= 128)
      ::rdm::launch<gather_max_kernel_body<32>, gather_max_kernel<32>, 256>(dim3(ceil_div<int64_t>(m, 8), 1), 0, st, x, ns, ci, lx, idx, mi, hi, li, width, y, ly, ord, i32);
    else if (m * ceil_div<int64_t>(c, 256) <= 4096)  // coarse levels: a workgroup per (query, chunk), slots split over its wavefronts
      ::rdm::launch<gather_max_kernel_body<64, true>, gather_max_kernel<64, true>, 256>(dim3(static_cast<unsigned>(m), ceil_div<int64_t>(c, 256)), 0, st, x, ns, ci, lx,
                         idx, mi, hi, li, width, y, ly, ord, i32);
    else
      ::rdm::launch<gather_max_kernel_body<64>, gather_max_kernel<64>, 256>(dim3(ceil_div<int64_t>(m, 4), ceil_div<int64_t>(c, 256)), 0, st, x, ns, ci, lx, idx, mi,
                         hi, li, width, y, ly, ord, i32);
  }
  return launch_status("gather_max_kernel");
}

extern "C" int rdm_gather_max(const float* x, int64_t n_s, int64_t c, int64_t ldx, const int64_t* idx,
                              int64_t m, int64_t h, int64_t ldi, const int32_t* width, float* y,
                              int64_t ldy, void* stream) {
  return rdm::gather_max_ordered(x, n_s, c, ldx, idx, m, h, ldi, width, y, ldy, nullptr, 0, stream);
}

// The same with 16-byte accesses (c1, c2 and every row stride multiples of 4, 16-byte aligned bases): one wavefront per row.
__device__ __forceinline__ void upsample_concat_vec_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* coarse, int n_coarse, int c1, int ld1,
                                                                  const int64_t* idx, int ldi, const float* skip, int c2, int ld2,
                                                                  int m_total, float* y, int ldy) {
  (void)blockIdx; (void)gridDim;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= m_total) return;
  const int lane = threadIdx.x & 63;
  const int64_t id = idx[static_cast<int64_t>(m) * ldi];
  const bool ok = id >= 0 && id < n_coarse;
  const float4* src1 = reinterpret_cast<const float4*>(coarse + (ok ? id : 0) * ld1);
  const float4* src2 = reinterpret_cast<const float4*>(skip + static_cast<int64_t>(m) * ld2);
  float4* dst = reinterpret_cast<float4*>(y + static_cast<int64_t>(m) * ldy);
  const int q1 = c1 / 4, q2 = c2 / 4, qy = ldy / 4;
  for (int q = lane; q < qy; q += 64) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < q1) {
      if (ok) v = src1[q];
    } else if (q < q1 + q2) {
      v = src2[q - q1];
    }
    dst[q] = v;
  }
}
__global__ __launch_bounds__(256) void upsample_concat_vec_kernel(const float* coarse, int n_coarse, int c1, int ld1,
                                                                  const int64_t* idx, int ldi, const float* skip, int c2, int ld2,
                                                                  int m_total, float* y, int ldy) { upsample_concat_vec_kernel_body(blockIdx, gridDim, coarse, n_coarse, c1, ld1, idx, ldi, skip, c2, ld2, m_total, y, ldy); }


extern "C" int rdm_upsample_concat(const float* coarse, int64_t n_coarse, int64_t c1, int64_t ld1,
                                   const int64_t* idx, int64_t ldi, const float* skip, int64_t c2,
                                   int64_t ld2, int64_t m, float* y, int64_t ldy, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(coarse && idx && skip && y, "rdm_upsample_concat: null pointer");
  RDM_REQUIRE(ldy >= c1 + c2, "rdm_upsample_concat: ldy too small");
  if (m == 0) return RDM_OK;
  const bool vec = c1 % 4 == 0 && c2 % 4 == 0 && ld1 % 4 == 0 && ld2 % 4 == 0 && ldy % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(coarse) | reinterpret_cast<uintptr_t>(skip) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  RDM_DUP_LOOP("ups")
  if (vec)
    ::rdm::launch<upsample_concat_vec_kernel_body, upsample_concat_vec_kernel, 256>(dim3(static_cast<unsigned>(ceil_div<int64_t>(m, 4))), 0, static_cast<hipStream_t>(stream), coarse, static_cast<int>(n_coarse), static_cast<int>(c1),
                       static_cast<int>(ld1), idx, static_cast<int>(ldi), skip, static_cast<int>(c2), static_cast<int>(ld2),
                       static_cast<int>(m), y, static_cast<int>(ldy));
  else
  ::rdm::launch<upsample_concat_kernel_body, upsample_concat_kernel, 256>(dim3(static_cast<unsigned>(m)), 0, static_cast<hipStream_t>(stream), coarse, static_cast<int>(n_coarse),
                     static_cast<int>(c1), static_cast<int>(ld1), idx, static_cast<int>(ldi), skip,
                     static_cast<int>(c2), static_cast<int>(ld2), static_cast<int>(m), y,
                     static_cast<int>(ldy));
  return launch_status("upsample_concat_kernel");
}
