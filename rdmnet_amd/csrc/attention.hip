// a7 -- 3DRoFormer attention: learned rotary embedding + fused softmax(QK^T/sqrt(d)) V.
//
// Reference: rdmnet/thdroformer/thdroformer.py:56-85 (RotaryPositionalEmbedding.forward),
// :20-40 (dynamic_attention, k=None => dense softmax), :88-139 (RPEMultiHeadAttention),
// geotransformer/modules/transformer/vanilla_transformer.py:51-66 (cross attention).
//
// rdm_rope: theta = 2*pi*sigmoid(emb) per (token, head, pair); (x0, x1) -> (x0 cos - x1 sin,
//   x1 cos + x0 sin) applied in place to q and k.
// rdm_attention: one workgroup = 16 queries of one head (d = 32), its four wavefronts split the
//   keys and merge their online-softmax partials through LDS.  S^T = K Q^T is formed with
//   v_mfma_f32_16x16x4_f32 (the contraction index is permuted so every lane reads 8 contiguous
//   floats of its key/query row), softmax runs online in registers (the 16 columns of the MFMA
//   result are the 16 queries, so a lane owns one query and 4 keys: the row reduction is 3 adds and
//   two cross-lane steps), and P is fed straight back as the A operand of the P V MFMAs -- no LDS, no
//   score matrix in memory.  It is MFMA work because it is a genuine dense contraction; at <= 450
//   tokens the op is latency- not throughput-bound.
#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "lockstep.h"

namespace {

using namespace rdm;

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kHeadDim = 32;

__device__ __forceinline__ void rope_body(const dim3 blockIdx, const dim3 gridDim, float* q, int ldq, float* k, int ldk,
                                          const float* emb, int lde, int n, int pairs) {
  (void)gridDim;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * pairs) return;
  const int row = i / pairs, p = i % pairs;
  const float e = emb[static_cast<int64_t>(row) * lde + p];
  const float sig = 1.0f / (1.0f + expf(-e));
  const float theta = sig * 3.14159265359f * 2.0f;
  const float c = cosf(theta), s = sinf(theta);
  float* a = q + static_cast<int64_t>(row) * ldq + 2 * p;
  const float q0 = a[0], q1 = a[1];
  a[0] = q0 * c + (-q1) * s;
  a[1] = q1 * c + q0 * s;
  if (k) {
    float* b = k + static_cast<int64_t>(row) * ldk + 2 * p;
    const float k0 = b[0], k1 = b[1];
    b[0] = k0 * c + (-k1) * s;
    b[1] = k1 * c + k0 * s;
  }
}
__global__ __launch_bounds__(256) void rope_kernel(float* q, int ldq, float* k, int ldk, const float* emb, int lde, int n, int pairs) {
  rope_body(blockIdx, gridDim, q, ldq, k, ldk, emb, lde, n, pairs);
}

struct AttnArgs {
  const float* q;
  const float* k;
  const float* v;
  float* out;
  int nq, nk, heads;
  int ldq, ldk, ldv, ldo;
  float inv_scale;  // sqrt(d)
  // optional second, independent segment in the same launch (self-attention of the second cloud): workgroups
  // blockIdx.x >= seg0_blocks take queries/keys starting at row `row1` with nq1 / nk1 rows; seg0_blocks = 0: none
  int seg0_blocks, row1, nq1, nk1;
};

// One workgroup = 16 queries of one head; its 4 wavefronts split the key tiles (tile % 4 == wave)
// and merge their online-softmax partials (m, l, O) through LDS at the end.
typedef short bf16x4_t __attribute__((ext_vector_type(4)));

// fp32 -> bf16 bits, round to nearest even (inputs are finite)
__device__ __forceinline__ short to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<short>(u >> 16);
}
__device__ __forceinline__ bf16x4_t pack_bf16(float a, float b, float c, float d) {
  bf16x4_t r;
  r[0] = to_bf16(a); r[1] = to_bf16(b); r[2] = to_bf16(c); r[3] = to_bf16(d);
  return r;
}

// BF16 = false: fp32 operands on the 16x16x4 f32 MFMA.  BF16 = true (configuration "bf16 attention"):
// Q, K, V and the probabilities are rounded to bf16 in registers and contracted on the 16x16x16 bf16 MFMA;
// logits, softmax statistics and both accumulators stay fp32.  HBM tensors are fp32 either way.
template <bool BF16>
__device__ __forceinline__ void attention_body(const dim3 blockIdx, const dim3 gridDim, AttnArgs a_in) {
  (void)gridDim;
  AttnArgs a = a_in;
  int bx = blockIdx.x;
  if (a.seg0_blocks > 0 && bx >= a.seg0_blocks) {  // block-uniform: the second cloud's rows
    bx -= a.seg0_blocks;
    a.q += static_cast<int64_t>(a.row1) * a.ldq;
    a.k += static_cast<int64_t>(a.row1) * a.ldk;
    a.v += static_cast<int64_t>(a.row1) * a.ldv;
    a.out += static_cast<int64_t>(a.row1) * a.ldo;
    a.nq = a.nq1;
    a.nk = a.nk1;
  }
  __shared__ float sm[4][16], sl[4][16];
  __shared__ float so[4][16][kHeadDim + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q0 = bx * 16;
  const int head = blockIdx.y;
  const int g = lane >> 4, x = lane & 15;
  const int hoff = head * kHeadDim;

  // Q fragment: query q0+x, features 8g .. 8g+7 of this head (contraction index = 8*kappa + step)
  float qf[8];
  {
    const int qi = min(q0 + x, a.nq - 1);
    const float4* p = reinterpret_cast<const float4*>(a.q + static_cast<int64_t>(qi) * a.ldq + hoff + 8 * g);
    const float4 u = p[0], w = p[1];
    qf[0] = u.x; qf[1] = u.y; qf[2] = u.z; qf[3] = u.w;
    qf[4] = w.x; qf[5] = w.y; qf[6] = w.z; qf[7] = w.w;
  }
  float m_run = -INFINITY, l_run = 0.f;   // per query x (replicated over g)
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};  // O[query 4g+r][d = 2x + t]

  // K / V fragments of a key tile: the next tile's loads are issued before the current tile is multiplied
  float kf_n[8];
  float2 vv_n[4];
  auto load_tile = [&](int k0) {
    const int ki = min(k0 + x, a.nk - 1);
    const float4* p = reinterpret_cast<const float4*>(a.k + static_cast<int64_t>(ki) * a.ldk + hoff + 8 * g);
    const float4 u = p[0], w = p[1];
    kf_n[0] = u.x; kf_n[1] = u.y; kf_n[2] = u.z; kf_n[3] = u.w;
    kf_n[4] = w.x; kf_n[5] = w.y; kf_n[6] = w.z; kf_n[7] = w.w;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int kj = min(k0 + 4 * g + st, a.nk - 1);
      vv_n[st] = *reinterpret_cast<const float2*>(a.v + static_cast<int64_t>(kj) * a.ldv + hoff + 2 * x);
    }
  };
  if (wave * 16 < a.nk) load_tile(wave * 16);
  for (int k0 = wave * 16; k0 < a.nk; k0 += 64) {
    // ---- S^T[key 4g'+r][query x] over this key tile
    float kf[8];
    float2 vv[4];
#pragma unroll
    for (int t = 0; t < 8; ++t) kf[t] = kf_n[t];
#pragma unroll
    for (int st = 0; st < 4; ++st) vv[st] = vv_n[st];
    if (k0 + 64 < a.nk) load_tile(k0 + 64);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BF16) {
      // contraction index = 8*kappa + 4*step + i: lane group kappa holds features 8*kappa .. 8*kappa+7
      s = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack_bf16(kf[0], kf[1], kf[2], kf[3]),
                                                    pack_bf16(qf[0], qf[1], qf[2], qf[3]), s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack_bf16(kf[4], kf[5], kf[6], kf[7]),
                                                    pack_bf16(qf[4], qf[5], qf[6], qf[7]), s, 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t], qf[t], s, 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[r] = (k0 + 4 * g + r < a.nk) ? s[r] / a.inv_scale : -INFINITY;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = expf(m_run - m_new);
    float psum = 0.f;
    f32x4 p;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = expf(s[r] - m_new);
      psum += p[r];
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // ---- rescale O rows (query 4g+r lives in lane 4g+r of group 0)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ar = __shfl(alpha, 4 * g + r, 64);
      o0[r] *= ar;
      o1[r] *= ar;
    }
    // ---- O += P V : step st uses key 4*kappa + st from lane group kappa, i.e. register st of p
    if constexpr (BF16) {
      // lane (x, kappa) of A holds P[query x][keys 4*kappa .. 4*kappa+3] = exactly the registers of p
      const bf16x4_t pa = pack_bf16(p[0], p[1], p[2], p[3]);
      o0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pack_bf16(vv[0].x, vv[1].x, vv[2].x, vv[3].x), o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pack_bf16(vv[0].y, vv[1].y, vv[2].y, vv[3].y), o1, 0, 0, 0);
    } else {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p[st], vv[st].x, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p[st], vv[st].y, o1, 0, 0, 0);
      }
    }
  }
  // ---- merge the four partial softmaxes
  if (g == 0) {
    sm[wave][x] = m_run;
    sl[wave][x] = l_run;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    so[wave][4 * g + r][2 * x] = o0[r];
    so[wave][4 * g + r][2 * x + 1] = o1[r];
  }
  __syncthreads();
  // thread t of the block finalises (query t>>4, features 2*(t&15), +1): 256 threads = 16 x 16 pairs
  {
    const int qq = threadIdx.x >> 4, dd = 2 * (threadIdx.x & 15);
    const int qi = q0 + qq;
    float mt = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mt = fmaxf(mt, sm[w][qq]);
    float lt = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = sl[w][qq] > 0.f ? expf(sm[w][qq] - mt) : 0.f;  // a wave without keys has l = 0
      lt += sl[w][qq] * f;
      a0 += so[w][qq][dd] * f;
      a1 += so[w][qq][dd + 1] * f;
    }
    if (qi < a.nq)
      *reinterpret_cast<float2*>(a.out + static_cast<int64_t>(qi) * a.ldo + hoff + dd) = make_float2(a0 / lt, a1 / lt);
  }
}
template <bool BF16>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) { attention_body<BF16>(blockIdx, gridDim, a); }

// vote.py:98-108: xyz + clamp(offset, -limit, +limit)
__device__ __forceinline__ void vote_shift_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* xyz, const float* off, int ldo, int n, float lx, float ly,
                                  float lz, float* out) {
  (void)blockIdx; (void)gridDim;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 3) return;
  const int row = i / 3, d = i % 3;
  const float lim = d == 0 ? lx : (d == 1 ? ly : lz);
  float o = off[static_cast<int64_t>(row) * ldo + d];
  o = o > lim ? lim : o;
  o = o < -lim ? -lim : o;
  out[i] = xyz[i] + o;
}
__global__ void vote_shift_kernel(const float* xyz, const float* off, int ldo, int n, float lx, float ly,
                                  float lz, float* out) { vote_shift_kernel_body(blockIdx, gridDim, xyz, off, ldo, n, lx, ly, lz, out); }


// sigmoid + clamp[0,1] of a strided column (model_infer.py:161-162, 171-172, 199-202)
__device__ __forceinline__ void sigmoid_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* x, int ldx, int n, float* out) {
  (void)blockIdx; (void)gridDim;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 1.0f / (1.0f + expf(-x[static_cast<int64_t>(i) * ldx]));
  out[i] = fminf(fmaxf(s, 0.f), 1.f);
}
__global__ void sigmoid_kernel(const float* x, int ldx, int n, float* out) { sigmoid_kernel_body(blockIdx, gridDim, x, ldx, n, out); }


// F.normalize(p=2, dim=1) (model_infer.py:248-249); one wavefront per row
__device__ __forceinline__ void l2_normalize_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* x, int ldx, int n, int c, float* y, int ldy) {
  (void)blockIdx; (void)gridDim;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < c; i += 64) {
    const float v = x[static_cast<int64_t>(row) * ldx + i];
    s += v * v;
  }
  const float nrm = fmaxf(__fsqrt_rn(wave_sum(s)), 1e-12f);
  for (int i = lane; i < c; i += 64) y[static_cast<int64_t>(row) * ldy + i] = x[static_cast<int64_t>(row) * ldx + i] / nrm;
}
__global__ void l2_normalize_kernel(const float* x, int ldx, int n, int c, float* y, int ldy) { l2_normalize_kernel_body(blockIdx, gridDim, x, ldx, n, c, y, ldy); }


}  // namespace

extern "C" int rdm_rope(float* q, int64_t ldq, float* k, int64_t ldk, const float* emb, int64_t lde,
                        int64_t n, int64_t d_model, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(q && emb && n >= 0 && d_model % 2 == 0, "rdm_rope: bad arguments");
  if (n == 0) return RDM_OK;
  const int pairs = static_cast<int>(d_model / 2);
  launch<rope_body, rope_kernel, 256>(dim3(ceil_div<int64_t>(n * pairs, 256)), 0, static_cast<hipStream_t>(stream), q,
                                      static_cast<int>(ldq), k, static_cast<int>(ldk), emb, static_cast<int>(lde), static_cast<int>(n), pairs);
  return launch_status("rope_kernel");
}

static int attention_launch(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            float* out, int64_t ldo, int64_t n_q, int64_t n_k, int heads, int head_dim, bool bf16,
                            void* stream) {
  using namespace rdm;
  RDM_REQUIRE(q && k && v && out, "rdm_attention: null pointer");
  RDM_REQUIRE(head_dim == kHeadDim, "rdm_attention: head_dim must be %d", kHeadDim);
  RDM_REQUIRE(n_q >= 0 && n_k > 0 && heads > 0, "rdm_attention: bad sizes");
  RDM_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 2 == 0 && ldo % 2 == 0, "rdm_attention: strides must be padded");
  if (n_q == 0) return RDM_OK;
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.out = out;
  a.nq = static_cast<int>(n_q); a.nk = static_cast<int>(n_k); a.heads = heads;
  a.ldq = static_cast<int>(ldq); a.ldk = static_cast<int>(ldk); a.ldv = static_cast<int>(ldv);
  a.ldo = static_cast<int>(ldo);
  a.inv_scale = sqrtf(static_cast<float>(head_dim));
  a.seg0_blocks = 0; a.row1 = 0; a.nq1 = 0; a.nk1 = 0;
  const dim3 grid(ceil_div<int64_t>(n_q, 16), heads);
  RDM_DUP_LOOP("attn")
  if (bf16)
    launch<attention_body<true>, attention_kernel<true>, 256>(grid, 0, static_cast<hipStream_t>(stream), a);
  else
    launch<attention_body<false>, attention_kernel<false>, 256>(grid, 0, static_cast<hipStream_t>(stream), a);
  return launch_status("attention_kernel");
}

extern "C" int rdm_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                             int64_t ldv, float* out, int64_t ldo, int64_t n_q, int64_t n_k, int heads,
                             int head_dim, void* stream) {
  return attention_launch(q, ldq, k, ldk, v, ldv, out, ldo, n_q, n_k, heads, head_dim, false, stream);
}

// Self-attention of two stacked clouds in one launch: rows [0, n0) attend to rows [0, n0), rows [n0, n0 + n1) to rows
// [n0, n0 + n1) (thdroformer.py:225-236 runs the self layer on ref and src separately; same results as two
// rdm_attention calls).
extern "C" int rdm_attention_self_pair(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                       float* out, int64_t ldo, int64_t n0, int64_t n1, int heads, int head_dim, int bf16,
                                       void* stream) {
  using namespace rdm;
  RDM_REQUIRE(q && k && v && out, "rdm_attention_self_pair: null pointer");
  RDM_REQUIRE(head_dim == kHeadDim, "rdm_attention: head_dim must be %d", kHeadDim);
  RDM_REQUIRE(n0 >= 0 && n1 >= 0 && heads > 0, "rdm_attention_self_pair: bad sizes");
  RDM_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 2 == 0 && ldo % 2 == 0, "rdm_attention: strides must be padded");
  if (n0 == 0 || n1 == 0) {  // one cloud is empty: the plain launch on the other
    if (n0 + n1 == 0) return RDM_OK;
    const int64_t n = n0 + n1;  // (the non-empty cloud starts at row 0 either way)
    return attention_launch(q, ldq, k, ldk, v, ldv, out, ldo, n, n, heads, head_dim, bf16 != 0, stream);
  }
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.out = out;
  a.nq = static_cast<int>(n0); a.nk = static_cast<int>(n0); a.heads = heads;
  a.ldq = static_cast<int>(ldq); a.ldk = static_cast<int>(ldk); a.ldv = static_cast<int>(ldv);
  a.ldo = static_cast<int>(ldo);
  a.inv_scale = sqrtf(static_cast<float>(head_dim));
  a.seg0_blocks = static_cast<int>(ceil_div<int64_t>(n0, 16));
  a.row1 = static_cast<int>(n0); a.nq1 = static_cast<int>(n1); a.nk1 = static_cast<int>(n1);
  const dim3 grid(a.seg0_blocks + ceil_div<int64_t>(n1, 16), heads);
  RDM_DUP_LOOP("attn")
  if (bf16)
    launch<attention_body<true>, attention_kernel<true>, 256>(grid, 0, static_cast<hipStream_t>(stream), a);
  else
    launch<attention_body<false>, attention_kernel<false>, 256>(grid, 0, static_cast<hipStream_t>(stream), a);
  return launch_status("attention_kernel");
}

extern "C" int rdm_attention_bf16(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                  int64_t ldv, float* out, int64_t ldo, int64_t n_q, int64_t n_k, int heads,
                                  int head_dim, void* stream) {
  return attention_launch(q, ldq, k, ldk, v, ldv, out, ldo, n_q, n_k, heads, head_dim, true, stream);
}

extern "C" int rdm_vote_shift(const float* xyz, const float* offsets, int64_t ldo, int64_t n, float lx,
                              float ly, float lz, float* out, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(xyz && offsets && out && n >= 0, "rdm_vote_shift: bad arguments");
  if (n == 0) return RDM_OK;
  ::rdm::launch<vote_shift_kernel_body, vote_shift_kernel, 256>(dim3(ceil_div<int64_t>(3 * n, 256)), 0, static_cast<hipStream_t>(stream), xyz, offsets, static_cast<int>(ldo), static_cast<int>(n), lx,
                     ly, lz, out);
  return launch_status("vote_shift_kernel");
}

extern "C" int rdm_sigmoid_column(const float* x, int64_t ldx, int64_t n, float* out, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(x && out && n >= 0, "rdm_sigmoid_column: bad arguments");
  if (n == 0) return RDM_OK;
  ::rdm::launch<sigmoid_kernel_body, sigmoid_kernel, 256>(dim3(ceil_div<int64_t>(n, 256)), 0, static_cast<hipStream_t>(stream), x, static_cast<int>(ldx), static_cast<int>(n), out);
  return launch_status("sigmoid_kernel");
}

extern "C" int rdm_l2_normalize(const float* x, int64_t ldx, int64_t n, int64_t c, float* y, int64_t ldy,
                                void* stream) {
  using namespace rdm;
  RDM_REQUIRE(x && y && n >= 0 && c > 0, "rdm_l2_normalize: bad arguments");
  if (n == 0) return RDM_OK;
  ::rdm::launch<l2_normalize_kernel_body, l2_normalize_kernel, 256>(dim3(ceil_div<int64_t>(n, 4)), 0, static_cast<hipStream_t>(stream), x, static_cast<int>(ldx), static_cast<int>(n),
                     static_cast<int>(c), y, static_cast<int>(ldy));
  return launch_status("l2_normalize_kernel");
}
