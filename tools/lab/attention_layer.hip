// a7 -- one application of a 3DRoFormer attention layer as ONE launch.
//
// Reference: rdmnet/thdroformer/thdroformer.py:142-202 (RPEAttentionLayer / RPETransformerLayer), :204-251
// (RPEConditionalTransformer: self layers on ref and src, cross layers ref <- src then src <- UPDATED ref),
// geotransformer/modules/transformer/vanilla_transformer.py:15-129, output_layer.py:6-21.
//
// A pair runs two transformers of eight layers on <= 450 superpoints per cloud: every op is a few microseconds of
// work, so what a layer costs is its NUMBER of dependent launches (projection, rotary embedding, attention, tail =
// 4-6 launches per layer; with four pairs in flight every launch of the chain costs ~1.6 us of throughput per pair,
// tools/exp_dup.sh).  Everything of a layer that is local to a block of 16 rows is one kernel here:
//
//   phase A  hid = softmax(q k^T / sqrt(32)) v            (4 heads x 2 key halves = 8 wavefronts, merged through LDS)
//   phase B  y   = LayerNorm(hid Wo^T + bo + x)
//            out = LayerNorm(relu(y W1^T + b1) W2^T + b2 + y)
//   phase C  up to two projections of the NEW rows for the layers that follow (the next self layer's q|k|v with its
//            rotary embedding, the next cross layer's q and k|v, or the transformer's output projection)
//
// so the only launch boundaries left are the true ones: attention needs the keys/values of ALL rows of a cloud.
// Per (self, cross) layer pair: 3 launches instead of 10.
// Phases A and B follow attention_kernel (attention.hip) and attention_tail128_kernel (gemm.hip): same operand
// layouts, same fixed summation orders within a wavefront.
#include "../../include/rdmnet_hip.h"
#include "common.h"

namespace {

using namespace rdm;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4_t __attribute__((ext_vector_type(4)));
constexpr int kHeadDim = 32;

__device__ __forceinline__ short to_bf16(float f) {  // round to nearest even (inputs are finite)
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<short>(u >> 16);
}
__device__ __forceinline__ bf16x4_t pack_bf16(float a, float b, float c, float d) {
  bf16x4_t r;
  r[0] = to_bf16(a); r[1] = to_bf16(b); r[2] = to_bf16(c); r[3] = to_bf16(d);
  return r;
}

struct Proj {
  const float* w;
  const float* bias;
  float* dst;
  int ncols, ldw, ldd, rope_cols, segments;
};
struct LayerArgs {
  const float* q;
  const float* x;
  float* out;
  int ldq, ldx, ldo;
  int seg_blocks0;  // workgroups of segment 0; the rest belong to segment 1
  int row0[2], nq[2], nk[2];
  const float* k[2];
  const float* v[2];
  int ldk[2], ldv[2];
  float inv_scale;  // sqrt(head_dim)
  int projections_only;
  const float *wo, *bo, *g1, *be1, *w1, *b1, *w2, *b2, *g2, *be2;
  int ldwo, ldw1, ldw2;
  float eps;
  int nproj;
  Proj proj[2];
  const float* emb;
  int lde;
};

template <bool BF16>
__global__ __launch_bounds__(512) void attention_layer_kernel(LayerArgs a) {
  __shared__ __attribute__((aligned(16))) float ys[16][132];  // y (first LayerNorm), later the new rows for phase C
  __shared__ __attribute__((aligned(16))) float zs[16][260];  // hid (attention output), later the FFN activations
  __shared__ float red[4][8][16];
  __shared__ float sm[8][16], sl[8][16];
  __shared__ float so[8][16][kHeadDim + 1];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kb = lane >> 4;
  int bx = blockIdx.x, seg = 0;
  if (bx >= a.seg_blocks0) {  // block-uniform
    bx -= a.seg_blocks0;
    seg = 1;
  }
  const int nrows = min(16, a.nq[seg] - bx * 16);  // valid rows of this block (>= 1)
  const int row_base = a.row0[seg] + bx * 16;      // stacked row of local row 0
  auto grow = [&](int local) { return row_base + min(local, nrows - 1); };  // clamped: always a valid row
  const int c = 16 * w + i;  // this lane's column of the 128-wide products

  float o[4];  // the new rows: rows 4kb + r of column c
  if (!a.projections_only) {
    // ---- phase A: attention.  Wavefront w = head (w & 3), key half (w >> 2): key tiles k0 = 16 half, +32, ...
    {
      const int head = w & 3, half = w >> 2, hoff = head * kHeadDim;
      const int g = kb, xq = i;
      const float* K = a.k[seg];
      const float* V = a.v[seg];
      const int ldk = a.ldk[seg], ldv = a.ldv[seg], nk = a.nk[seg];
      float qf[8];
      {
        const float4* p = reinterpret_cast<const float4*>(a.q + static_cast<int64_t>(grow(xq)) * a.ldq + hoff + 8 * g);
        const float4 u = p[0], t = p[1];
        qf[0] = u.x; qf[1] = u.y; qf[2] = u.z; qf[3] = u.w;
        qf[4] = t.x; qf[5] = t.y; qf[6] = t.z; qf[7] = t.w;
      }
      float m_run = -INFINITY, l_run = 0.f;
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
      float kf_n[8];
      float2 vv_n[4];
      auto load_tile = [&](int k0) {
        const int ki = min(k0 + xq, nk - 1);
        const float4* p = reinterpret_cast<const float4*>(K + static_cast<int64_t>(ki) * ldk + hoff + 8 * g);
        const float4 u = p[0], t = p[1];
        kf_n[0] = u.x; kf_n[1] = u.y; kf_n[2] = u.z; kf_n[3] = u.w;
        kf_n[4] = t.x; kf_n[5] = t.y; kf_n[6] = t.z; kf_n[7] = t.w;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const int kj = min(k0 + 4 * g + st, nk - 1);
          vv_n[st] = *reinterpret_cast<const float2*>(V + static_cast<int64_t>(kj) * ldv + hoff + 2 * xq);
        }
      };
      if (half * 16 < nk) load_tile(half * 16);
      for (int k0 = half * 16; k0 < nk; k0 += 32) {
        float kf[8];
        float2 vv[4];
#pragma unroll
        for (int t = 0; t < 8; ++t) kf[t] = kf_n[t];
#pragma unroll
        for (int st = 0; st < 4; ++st) vv[st] = vv_n[st];
        if (k0 + 32 < nk) load_tile(k0 + 32);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BF16) {
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
          s[r] = (k0 + 4 * g + r < nk) ? s[r] / a.inv_scale : -INFINITY;
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
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ar = __shfl(alpha, 4 * g + r, 64);
          o0[r] *= ar;
          o1[r] *= ar;
        }
        if constexpr (BF16) {
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
      if (g == 0) {
        sm[w][xq] = m_run;
        sl[w][xq] = l_run;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        so[w][4 * g + r][2 * xq] = o0[r];
        so[w][4 * g + r][2 * xq + 1] = o1[r];
      }
    }
    __syncthreads();
    {  // merge the two key halves of every head (fixed order) into the 16 x 128 attention rows
      const int qq = tid >> 5, c4 = (tid & 31) * 4;
      const int hd = c4 >> 5, d0 = c4 & 31;
      const float m0 = sm[hd][qq], m1 = sm[hd + 4][qq], l0 = sl[hd][qq], l1 = sl[hd + 4][qq];
      const float mt = fmaxf(m0, m1);
      const float f0 = l0 > 0.f ? expf(m0 - mt) : 0.f, f1 = l1 > 0.f ? expf(m1 - mt) : 0.f;  // a half without keys has l = 0
      const float lt = l0 * f0 + l1 * f1;
      float4 hv;
      hv.x = (so[hd][qq][d0] * f0 + so[hd + 4][qq][d0] * f1) / lt;
      hv.y = (so[hd][qq][d0 + 1] * f0 + so[hd + 4][qq][d0 + 1] * f1) / lt;
      hv.z = (so[hd][qq][d0 + 2] * f0 + so[hd + 4][qq][d0 + 2] * f1) / lt;
      hv.w = (so[hd][qq][d0 + 3] * f0 + so[hd + 4][qq][d0 + 3] * f1) / lt;
      *reinterpret_cast<float4*>(&zs[qq][c4]) = hv;
    }

    // ---- phase B: the tail (attention_tail128_kernel with the attention rows already in LDS)
    float4 bw0[8], bw1[2][8], bw2[16];
    float xres[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xres[r] = a.x[static_cast<int64_t>(grow(4 * kb + r)) * a.ldx + c];
    const float bo = a.bo ? a.bo[c] : 0.f, b2 = a.b2 ? a.b2[c] : 0.f;
    const float b10 = a.b1 ? a.b1[32 * w + i] : 0.f, b11 = a.b1 ? a.b1[32 * w + 16 + i] : 0.f;
    const float g1 = a.g1[c], be1 = a.be1[c], g2 = a.g2[c], be2 = a.be2[c];
    const float* p1a = a.w1 + static_cast<int64_t>(32 * w + i) * a.ldw1 + 4 * kb;
    const float* p1b = p1a + 16ll * a.ldw1;
    const float* p2 = a.w2 + static_cast<int64_t>(c) * a.ldw2 + 4 * kb;
    {
      const float* p0 = a.wo + static_cast<int64_t>(c) * a.ldwo + 4 * kb;
#pragma unroll
      for (int s = 0; s < 8; ++s) bw0[s] = *reinterpret_cast<const float4*>(p0 + 16 * s);
    }
    __builtin_amdgcn_sched_barrier(0);

    auto layer_norm = [&](float (&v)[4], int slot, float gam, float bet) {
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = v[r];
      float mean[4];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = row16_sum(p[r]);
        if (i == 0)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[slot + pass][w][4 * kb + r] = p[r];
        lds_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = 0.f;
#pragma unroll
          for (int ww = 0; ww < 8; ++ww) t += red[slot + pass][ww][4 * kb + r];  // fixed order
          if (pass == 0) {
            mean[r] = t / 128.f;
            const float d = v[r] - mean[r];
            p[r] = d * d;
          } else {
            const float rstd = 1.0f / __fsqrt_rn(t / 128.f + a.eps);
            v[r] = (v[r] - mean[r]) * rstd * gam + bet;
          }
        }
      }
    };

    lds_barrier();  // zs = attention rows
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      bw1[0][s] = *reinterpret_cast<const float4*>(p1a + 16 * s);
      bw1[1][s] = *reinterpret_cast<const float4*>(p1b + 16 * s);
      const float4 ha = *reinterpret_cast<const float4*>(&zs[i][16 * s + 4 * kb]);
      const float at[4] = {ha.x, ha.y, ha.z, ha.w}, bt[4] = {bw0[s].x, bw0[s].y, bw0[s].z, bw0[s].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], bt[t], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) bw2[s] = *reinterpret_cast<const float4*>(p2 + 16 * s);
    __builtin_amdgcn_sched_barrier(0);
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = acc[r] + bo + xres[r];
    layer_norm(y, 0, g1, be1);
#pragma unroll
    for (int r = 0; r < 4; ++r) ys[4 * kb + r][c] = y[r];
    lds_barrier();

    f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      bw2[8 + s] = *reinterpret_cast<const float4*>(p2 + 16 * (8 + s));
      const float4 ya = *reinterpret_cast<const float4*>(&ys[i][16 * s + 4 * kb]);
      const float at[4] = {ya.x, ya.y, ya.z, ya.w};
      const float u0[4] = {bw1[0][s].x, bw1[0][s].y, bw1[0][s].z, bw1[0][s].w};
      const float u1[4] = {bw1[1][s].x, bw1[1][s].y, bw1[1][s].z, bw1[1][s].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], u0[t], z0, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], u1[t], z1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e0 = z0[r] + b10, e1 = z1[r] + b11;
      zs[4 * kb + r][32 * w + i] = e0 > 0.f ? e0 : 0.f;
      zs[4 * kb + r][32 * w + 16 + i] = e1 > 0.f ? e1 : 0.f;
    }
    lds_barrier();  // (every wavefront is past its reads of ys, too)

    acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float4 za = *reinterpret_cast<const float4*>(&zs[i][16 * s + 4 * kb]);
      const float at[4] = {za.x, za.y, za.z, za.w}, bt[4] = {bw2[s].x, bw2[s].y, bw2[s].z, bw2[s].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], bt[t], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = acc[r] + b2 + y[r];
    layer_norm(o, 2, g2, be2);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * kb + r < nrows) a.out[static_cast<int64_t>(row_base + 4 * kb + r) * a.ldo + c] = o[r];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = a.out[static_cast<int64_t>(grow(4 * kb + r)) * a.ldo + c];
  }
  if (a.nproj == 0) return;

  // ---- phase C: projections of the new rows (W as nn.Linear stores it, [ncols, 128]; 128 columns per pass, wavefront w
  // the columns 16w .. 16w+15 of the pass; the next pass's weights are requested before this pass's MFMAs)
#pragma unroll
  for (int r = 0; r < 4; ++r) ys[4 * kb + r][c] = o[r];
  lds_barrier();
  float4 av[8];  // the A operand (the 16 new rows) is the same for every pass
#pragma unroll
  for (int s = 0; s < 8; ++s) av[s] = *reinterpret_cast<const float4*>(&ys[i][16 * s + 4 * kb]);
  float cs[4] = {1.f, 1.f, 1.f, 1.f}, sn[4] = {0.f, 0.f, 0.f, 0.f};
  bool have_cs = false;
  for (int pi = 0; pi < a.nproj; ++pi) {
    const Proj P = a.proj[pi];
    if (!((P.segments >> seg) & 1)) continue;  // block-uniform
    auto load_w = [&](float4 (&b)[8], int c0) {
      const float* wrow = P.w + static_cast<int64_t>(c0 + c) * P.ldw + 4 * kb;
#pragma unroll
      for (int s = 0; s < 8; ++s) b[s] = *reinterpret_cast<const float4*>(wrow + 16 * s);
    };
    auto pass = [&](const float4 (&b)[8], int c0) {
      const int col = c0 + c;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float at[4] = {av[s].x, av[s].y, av[s].z, av[s].w}, bt[4] = {b[s].x, b[s].y, b[s].z, b[s].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], bt[t], acc, 0, 0, 0);
      }
      const float bias = P.bias ? P.bias[col] : 0.f;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[r] + bias;
      if (c0 < P.rope_cols) {  // pass-uniform: learned rotary embedding (thdroformer.py:56-85), pair p = columns (2p, 2p+1)
        if (!have_cs) {
          const int p = c >> 1;  // q pass and k pass: the same pair index for the same lane
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = a.emb[static_cast<int64_t>(grow(4 * kb + r)) * a.lde + p];
            const float sig = 1.0f / (1.0f + expf(-e));
            const float theta = sig * 3.14159265359f * 2.0f;
            cs[r] = cosf(theta);
            sn[r] = sinf(theta);
          }
          have_cs = true;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float other = __shfl_xor(v[r], 1, 64);
          // even column x0 -> x0 cos - x1 sin, odd column x1 -> x1 cos + x0 sin
          v[r] = v[r] * cs[r] + ((c & 1) ? other : -other) * sn[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * kb + r < nrows) P.dst[static_cast<int64_t>(row_base + 4 * kb + r) * P.ldd + col] = v[r];
    };
    float4 ba[8], bb[8];
    load_w(ba, 0);
    for (int c0 = 0; c0 < P.ncols; c0 += 256) {
      if (c0 + 128 < P.ncols) load_w(bb, c0 + 128);
      pass(ba, c0);
      if (c0 + 128 < P.ncols) {
        if (c0 + 256 < P.ncols) load_w(ba, c0 + 256);
        pass(bb, c0 + 128);
      }
    }
  }
}

}  // namespace

extern "C" int rdm_attention_layer(const rdm_attention_layer_args* in, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(in, "rdm_attention_layer: null arguments");
  RDM_REQUIRE(in->heads == 4 && in->head_dim == kHeadDim, "rdm_attention_layer: supports 4 heads x 32 (got %d x %d)", in->heads,
              in->head_dim);
  RDM_REQUIRE(in->n_segments == 1 || in->n_segments == 2, "rdm_attention_layer: one or two segments");
  RDM_REQUIRE(in->out && in->ldo >= 128, "rdm_attention_layer: out missing");
  RDM_REQUIRE(in->n_projections >= 0 && in->n_projections <= 2, "rdm_attention_layer: at most two projections");
  LayerArgs a;
  std::memset(&a, 0, sizeof(a));
  a.q = in->q; a.x = in->x; a.out = in->out;
  a.ldq = static_cast<int>(in->ldq); a.ldx = static_cast<int>(in->ldx); a.ldo = static_cast<int>(in->ldo);
  int64_t blocks = 0;
  for (int s = 0; s < in->n_segments; ++s) {
    RDM_REQUIRE(in->n_q[s] >= 0 && in->row0[s] >= 0, "rdm_attention_layer: bad segment %d", s);
    a.row0[s] = static_cast<int>(in->row0[s]);
    a.nq[s] = static_cast<int>(in->n_q[s]);
    if (s == 0) a.seg_blocks0 = static_cast<int>(ceil_div<int64_t>(in->n_q[0], 16));
    blocks += ceil_div<int64_t>(in->n_q[s], 16);
  }
  a.projections_only = in->projections_only != 0;
  if (!a.projections_only) {
    RDM_REQUIRE(in->q && in->x && in->ldq % 4 == 0 && in->ldq >= 128 && in->ldx >= 128, "rdm_attention_layer: q / x missing");
    RDM_REQUIRE((reinterpret_cast<uintptr_t>(in->q) & 15) == 0, "rdm_attention_layer: q must be 16-byte aligned");
    for (int s = 0; s < in->n_segments; ++s) {
      if (in->n_q[s] == 0) continue;
      RDM_REQUIRE(in->k[s] && in->v[s] && in->n_k[s] > 0 && in->ldk[s] % 4 == 0 && in->ldv[s] % 2 == 0,
                  "rdm_attention_layer: keys / values of segment %d missing", s);
      RDM_REQUIRE((reinterpret_cast<uintptr_t>(in->k[s]) & 15) == 0 && (reinterpret_cast<uintptr_t>(in->v[s]) & 7) == 0,
                  "rdm_attention_layer: k must be 16-byte, v 8-byte aligned");
      a.k[s] = in->k[s]; a.v[s] = in->v[s];
      a.ldk[s] = static_cast<int>(in->ldk[s]); a.ldv[s] = static_cast<int>(in->ldv[s]); a.nk[s] = static_cast<int>(in->n_k[s]);
    }
    RDM_REQUIRE(in->wo && in->w1 && in->w2 && in->gamma1 && in->beta1 && in->gamma2 && in->beta2,
                "rdm_attention_layer: tail weights missing");
    RDM_REQUIRE(in->ld_wo % 4 == 0 && in->ld_w1 % 4 == 0 && in->ld_w2 % 4 == 0 && in->ld_wo >= 128 && in->ld_w1 >= 128 &&
                    in->ld_w2 >= 256,
                "rdm_attention_layer: bad weight strides");
    RDM_REQUIRE(((reinterpret_cast<uintptr_t>(in->wo) | reinterpret_cast<uintptr_t>(in->w1) | reinterpret_cast<uintptr_t>(in->w2)) & 15) == 0,
                "rdm_attention_layer: weights must be 16-byte aligned");
    a.wo = in->wo; a.bo = in->bo; a.g1 = in->gamma1; a.be1 = in->beta1; a.w1 = in->w1; a.b1 = in->b1; a.w2 = in->w2;
    a.b2 = in->b2; a.g2 = in->gamma2; a.be2 = in->beta2;
    a.ldwo = static_cast<int>(in->ld_wo); a.ldw1 = static_cast<int>(in->ld_w1); a.ldw2 = static_cast<int>(in->ld_w2);
  }
  a.inv_scale = sqrtf(static_cast<float>(in->head_dim));
  a.eps = in->eps;
  a.nproj = in->n_projections;
  for (int p = 0; p < in->n_projections; ++p) {
    const rdm_layer_projection& P = in->proj[p];
    RDM_REQUIRE(P.w && P.dst && P.ncols > 0 && P.ncols % 128 == 0 && P.ldw >= 128 && P.ldw % 4 == 0 && P.ldd >= P.ncols,
                "rdm_attention_layer: projection %d: bad shape (ncols %d)", p, P.ncols);
    RDM_REQUIRE((reinterpret_cast<uintptr_t>(P.w) & 15) == 0, "rdm_attention_layer: projection weights must be 16-byte aligned");
    RDM_REQUIRE(P.rope_cols >= 0 && P.rope_cols % 128 == 0 && P.rope_cols <= P.ncols && (P.rope_cols == 0 || in->emb),
                "rdm_attention_layer: projection %d: bad rope_cols %d", p, P.rope_cols);
    a.proj[p].w = P.w; a.proj[p].bias = P.bias; a.proj[p].dst = P.dst; a.proj[p].ncols = P.ncols; a.proj[p].ldw = P.ldw;
    a.proj[p].ldd = P.ldd; a.proj[p].rope_cols = P.rope_cols; a.proj[p].segments = P.segments;
  }
  a.emb = in->emb;
  a.lde = static_cast<int>(in->lde);
  if (blocks == 0) return RDM_OK;
  const dim3 grid(static_cast<unsigned>(blocks));
  if (in->bf16)
    hipLaunchKernelGGL(attention_layer_kernel<true>, grid, dim3(512), 0, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(attention_layer_kernel<false>, grid, dim3(512), 0, static_cast<hipStream_t>(stream), a);
  return launch_status("attention_layer_kernel");
}
