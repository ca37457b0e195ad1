// a4 -- KPConv neighbourhood aggregation (the gather-bound half of the rigid kernel-point convolution).
//
// Reference: geotransformer/modules/kpconv/kpconv.py:79-122.  For every query m:
//   rel_h   = s_points[idx[m,h]] - q_points[m]                  (pad index -> point at +1e6)   :91-93
//   w[h,k]  = max(0, 1 - sqrt(|rel_h - kp_k|^2) / sigma)                                       :96-99
//   WF[m, k, c] = sum_h w[h,k] * s_feats[idx[m,h], c]           (pad index -> zero row)        :103-105
//   nn[m]   = max(1, #{h : sum_c s_feats[idx[m,h], c] > 0})                                    :113-115
// The second half, out = (WF viewed [M, 15*C]) x W[15*C, C'] / nn + bias (:107-121), is a dense GEMM
// and runs in rdm_gemm with `rowdiv = nn`.
//
// Mapping: one wavefront per query.  Phase 1 loads the index row coalesced and stages the relative
// neighbour positions in LDS.  Phase 2 walks the neighbours four at a time; the 15 kernel-point
// influences of those four neighbours ARE the A operand of v_mfma_f32_16x16x4_f32 (lane = (neighbour
// g, kernel point j)) and the gathered feature rows are the B operand, fetched as 8/16-byte vectors
// (16 lanes x 16 B = one contiguous 256-B piece of a feature row, four rows per instruction).  The
// channel -> tile assignment is permuted (channel = 16*VEC*u + VEC*j + e) so that both the gathers and
// the WF stores are vector accesses.
#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "internal.h"
#include "lockstep.h"

namespace {

using namespace rdm;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKP = 15;          // kernel points (cfg.backbone.kernel_size)
constexpr int kMaxH = 128;       // neighbour slots staged in LDS at a time (wider rows run in chunks)
constexpr int kWaves = 4;

struct KpArgs {
  const float* q_points;   // [M,3]
  const float* s_points;   // [Ns,3]
  const float* s_feats;    // [Ns, ldf]
  const unsigned char* s_pos;  // [Ns] 1 iff sum_c feats > 0
  const int64_t* idx;      // [M, ldi] (int32 elements when i32)
  int i32;
  const float* kp;         // [15,3]
  const int32_t* width;    // optional device int: effective row width (min(limit, max_count))
  const float4* order;     // optional processing order: query row = int bits of order[unit].w
  int units_per_block;     // work units (query x channel slice) per workgroup
  int split;               // channel slices per query when the kernel's SPLIT parameter is 0 (generic channel counts)
  int xcd_remap;           // 1: workgroups re-mapped so that each XCD owns a contiguous range of the (cell-ordered) units
  float* wf;               // [M, ldw] (>= 15*C)
  float* nn;               // [M]
  int M, Ns, H, C;
  int ldf, ldi, ldw;
  float sigma;
};

// VEC x U x 16 channels per wavefront; SPLIT wavefronts share one query (channel slices) so that the
// coarse levels (few hundred queries, 256-512 channels) still fill the chip; PF neighbour groups of
// four are fetched before the first is consumed (the gather -> MFMA chain is latency-bound otherwise).
template <int VEC, int U, int SPLIT, int PF>
__device__ __forceinline__ void kpconv_gather_kernel_body(const dim3 blockIdx, const dim3 gridDim, KpArgs a) {
  (void)blockIdx; (void)gridDim;
  __shared__ float4 nb[kWaves][kMaxH];  // rel.xyz, w = bit pattern of the support row (or -1)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // Work distribution.  Large levels: a workgroup owns `qpb` consecutive work units (= queries in the
  // processing order, spatially coherent when `order` is given) and its wavefronts interleave over them,
  // so lines gathered for one query are still in this CU's L1 for its neighbours; workgroups are
  // re-mapped so that each XCD (own L2) gets a contiguous range of the level.  Small levels: one unit per
  // wavefront for parallelism.
  const int split = SPLIT > 0 ? SPLIT : a.split;  // (SPLIT = 0: any multiple of 16*VEC*U channels, slices counted at run time)
  const int total_units = a.M * split;
  const int qpb = a.units_per_block;
  const int nblk = gridDim.x;
  int blk = blockIdx.x;
  if ((qpb > kWaves || a.xcd_remap) && nblk >= 16) {  // bijective XCD remap (8 XCDs, round-robin dispatch)
    const int q = nblk / 8, rr = nblk % 8, xcd = blk % 8, within = blk / 8;
    blk = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + within;
  }
  const int g = lane >> 4, j = lane & 15;
  const float kx = j < kKP ? a.kp[3 * j] : 0.f, ky = j < kKP ? a.kp[3 * j + 1] : 0.f,
              kz = j < kKP ? a.kp[3 * j + 2] : 0.f;
  // hardware sqrt and a reciprocal multiply (1 ulp each) instead of the correctly rounded sqrt/divide:
  // the kernel is VALU-issue bound (15 influences per neighbour), and float outputs are compared with a
  // tolerance anyway
  const float inv_sigma = 1.0f / a.sigma;
  for (int u0 = wave; u0 < qpb; u0 += kWaves) {
  const int unit = blk * qpb + u0;
  if (unit >= total_units) break;
  const int slice = unit % split;
  const int m = a.order ? __float_as_int(a.order[unit / split].w) : unit / split;
  int H = a.H;
  if (a.width) H = min(H, *a.width);
  const int c_base = slice * (16 * VEC * U);  // first channel of this wavefront's slice

  const float qx = a.q_points[3 * m], qy = a.q_points[3 * m + 1], qz = a.q_points[3 * m + 2];
  int positives = 0;
  f32x4 acc[VEC * U];
#pragma unroll
  for (int t = 0; t < VEC * U; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // neighbour slots in chunks of the LDS staging row (kMaxH = 128 slots: one chunk for every KITTI limit; the reference
  // takes whatever calibrate_neighbors_stack_mode returns, utils/data.py:195-220)
  for (int hc = 0; hc < H; hc += kMaxH) {
  const int Hc = min(H - hc, kMaxH);
  // ---- phase 1: neighbour rows, relative positions, positive-row count
  int Hq = 0;  // slots up to the last real neighbour: shadow neighbours contribute exact zeros, and the searches pad at the end
  for (int hb = 0; hb < Hc; hb += 64) {  // (wavefront-uniform trip count: Hq must be the same in every lane)
    const int h = hb + lane;
    const int64_t id = h < Hc ? ld_index(a.idx, static_cast<int64_t>(m) * a.ldi + hc + h, a.i32) : -1;
    const bool real = id >= 0 && id < a.Ns;
    const unsigned long long rm = __builtin_amdgcn_ballot_w64(real);
    if (rm) Hq = hb + 64 - __builtin_clzll(rm);
    float4 v;
    if (real) {
      v.x = a.s_points[3 * id] - qx;
      v.y = a.s_points[3 * id + 1] - qy;
      v.z = a.s_points[3 * id + 2] - qz;
      v.w = __int_as_float(static_cast<int>(id));
      positives += a.s_pos[id];
    } else {  // shadow neighbour: point at 1e6, zero features
      v.x = 1.0e6f - qx;
      v.y = 1.0e6f - qy;
      v.z = 1.0e6f - qz;
      v.w = __int_as_float(-1);
    }
    if (h < Hc) nb[wave][h] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- phase 2: PF groups of four neighbours per trip
  for (int h0 = 0; h0 < Hq; h0 += 4 * PF) {
    float w[PF];
    float f[PF][U][VEC];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int h = h0 + 4 * p + g;
      int id = -1;
      w[p] = 0.f;
      if (h < Hq) {
        const float4 v = nb[wave][h];
        id = __float_as_int(v.w);
        const float dx = v.x - kx, dy = v.y - ky, dz = v.z - kz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        w[p] = fmaxf(0.f, 1.f - __builtin_amdgcn_sqrtf(d2) * inv_sigma);
        if (j >= kKP || id < 0) w[p] = 0.f;
      }
      const float* row = a.s_feats + static_cast<int64_t>(id < 0 ? 0 : id) * a.ldf + c_base + VEC * j;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (id >= 0) {
          if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(row + 16 * VEC * u);
            f[p][u][0] = t.x; f[p][u][1] = t.y; f[p][u][2] = t.z; f[p][u][VEC - 1] = t.w;
          } else if (VEC == 2) {
            const float2 t = *reinterpret_cast<const float2*>(row + 16 * VEC * u);
            f[p][u][0] = t.x; f[p][u][VEC - 1] = t.y;
          } else {
            f[p][u][0] = row[16 * VEC * u];
          }
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) f[p][u][e] = 0.f;
        }
      }
    }
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          acc[u * VEC + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p], f[p][u][e], acc[u * VEC + e], 0, 0, 0);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // nb[wave] is rewritten by the next chunk / unit
  __builtin_amdgcn_wave_barrier();
  }  // chunk loop
  positives = wave_sum_i(positives);

  // ---- store WF[m, k, c]: accumulator row = 4*g + r = kernel point, column j
  float* out = a.wf + static_cast<int64_t>(m) * a.ldw + c_base;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = 4 * g + r;
    if (k >= kKP) continue;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float* p = out + k * a.C + 16 * VEC * u + VEC * j;
      if (VEC == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(acc[u * 4 + 0][r], acc[u * 4 + 1][r],
                                                    acc[u * 4 + 2][r], acc[u * 4 + 3][r]);
      } else if (VEC == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(acc[u * 2 + 0][r], acc[u * 2 + 1][r]);
      } else {
        p[0] = acc[u][r];
      }
    }
  }
  if (lane == 0 && slice == 0) a.nn[m] = static_cast<float>(positives > 1 ? positives : 1);
  }  // unit loop
}
template <int VEC, int U, int SPLIT, int PF>
__global__ __launch_bounds__(64 * kWaves) void kpconv_gather_kernel(KpArgs a) { kpconv_gather_kernel_body<VEC, U, SPLIT, PF>(blockIdx, gridDim, a); }


// First layer (C_in = 1, features == 1 for every real point, reference dataset.py:187-188 and
// model_infer.py:113): WF[m,k] = sum_h w[h,k] * f[idx], no matrix core needed.
__device__ __forceinline__ void kpconv_gather_c1_kernel_body(const dim3 blockIdx, const dim3 gridDim, KpArgs a) {
  (void)blockIdx; (void)gridDim;
  // one wavefront per query: phase 1 stages (relative position, feature) of every neighbour in LDS with
  // coalesced index reads; phase 2: lane (g, j) walks neighbours g, g+4, ... for kernel point j
  __shared__ float4 nb[kWaves][kMaxH];  // rel.xyz, w = feature (0 for shadow neighbours)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mu = blockIdx.x * kWaves + wave;
  if (mu >= a.M) return;
  const int m = a.order ? __float_as_int(a.order[mu].w) : mu;
  int H = a.H;
  if (a.width) H = min(H, *a.width);
  const float qx = a.q_points[3 * m], qy = a.q_points[3 * m + 1], qz = a.q_points[3 * m + 2];
  const int g = lane >> 4, j = lane & 15;
  const float kx = j < kKP ? a.kp[3 * j] : 0.f, ky = j < kKP ? a.kp[3 * j + 1] : 0.f,
              kz = j < kKP ? a.kp[3 * j + 2] : 0.f;
  const float inv_sigma = 1.0f / a.sigma;
  int positives = 0;
  float acc = 0.f;
  for (int hc = 0; hc < H; hc += kMaxH) {  // chunks of the staging row (one for every KITTI limit)
    const int Hc = min(H - hc, kMaxH);
    int Hq = 0;  // slots up to the last real neighbour (the rest are shadow neighbours: zero feature)
    for (int hb = 0; hb < Hc; hb += 64) {  // (wavefront-uniform trip count: Hq must be the same in every lane)
      const int h = hb + lane;
      const int64_t id = h < Hc ? ld_index(a.idx, static_cast<int64_t>(m) * a.ldi + hc + h, a.i32) : -1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const unsigned long long rm = __builtin_amdgcn_ballot_w64(id >= 0 && id < a.Ns);
      if (rm) Hq = hb + 64 - __builtin_clzll(rm);
      if (id >= 0 && id < a.Ns) {
        v.x = a.s_points[3 * id] - qx;
        v.y = a.s_points[3 * id + 1] - qy;
        v.z = a.s_points[3 * id + 2] - qz;
        v.w = a.s_feats[id * a.ldf];
        positives += a.s_pos[id];
      }
      if (h < Hc) nb[wave][h] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int h = g; h < Hq; h += 4) {
      const float4 v = nb[wave][h];
      const float dx = v.x - kx, dy = v.y - ky, dz = v.z - kz;
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      acc += fmaxf(0.f, 1.f - __builtin_amdgcn_sqrtf(d2) * inv_sigma) * v.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  positives = wave_sum_i(positives);
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  float* out = a.wf + static_cast<int64_t>(m) * a.ldw;
  if (g == 0) {
    if (j < kKP) out[j] = acc;
    else {
      for (int k = kKP; k < a.ldw; ++k) out[k] = 0.f;
      a.nn[m] = static_cast<float>(positives > 1 ? positives : 1);
    }
  }
}
__global__ __launch_bounds__(64 * kWaves) void kpconv_gather_c1_kernel(KpArgs a) { kpconv_gather_c1_kernel_body(blockIdx, gridDim, a); }


// 1 iff the row sum is positive (kpconv.py:113-114); one wavefront per row, fixed summation order.
__device__ __forceinline__ void row_positive_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* x, int n, int c, int ld, unsigned char* out) {
  (void)blockIdx; (void)gridDim;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < c; i += 64) s += x[static_cast<int64_t>(row) * ld + i];
  s = wave_sum(s);
  if (lane == 0) out[row] = s > 0.f ? 1 : 0;
}
__global__ void row_positive_kernel(const float* x, int n, int c, int ld, unsigned char* out) { row_positive_kernel_body(blockIdx, gridDim, x, n, c, ld, out); }


}  // namespace

extern "C" int rdm_row_positive(const float* x, int64_t n, int64_t c, int64_t ld, uint8_t* out,
                                void* stream) {
  using namespace rdm;
  RDM_REQUIRE(x && out && n >= 0 && c > 0, "rdm_row_positive: bad arguments");
  if (n == 0) return RDM_OK;
  ::rdm::launch<row_positive_kernel_body, row_positive_kernel, 256>(dim3(ceil_div<int64_t>(n, 4)), 0, static_cast<hipStream_t>(stream), x, static_cast<int>(n), static_cast<int>(c),
                     static_cast<int>(ld), out);
  return launch_status("row_positive_kernel");
}

namespace {
// Where the LDS-tile form of the gather beats the per-neighbour form (measured per shape on MI355X, tools/kpconv_bench.py,
// profiles/r05_kpconv_bench.txt): the tile form runs blocks-of-16 x slices workgroups, two per CU.
bool tile_gather_pays(int64_t c, int64_t m) {
  (void)c; (void)m;
  return false;  // (set from the measurements)
}
}  // namespace

int rdm::kpconv_gather_impl(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats, int64_t c,
                            int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                            const float* kernel_points, float sigma, float* wf, int64_t ldw, float* nn, const float* order_records,
                            int i32, void* stream, int form) {
  using namespace rdm;
  RDM_REQUIRE(q_points && s_points && s_feats && s_positive && idx && kernel_points && wf && nn,
              "rdm_kpconv_gather: null pointer");
  RDM_REQUIRE(m >= 0 && n_s > 0 && h > 0, "rdm_kpconv_gather: bad sizes (h=%lld)", (long long)h);
  RDM_REQUIRE(c == 1 || c % 32 == 0, "rdm_kpconv_gather: unsupported channel count %lld (1 or a multiple of 32)", (long long)c);
  RDM_REQUIRE(ldw >= kKP * c && (c == 1 || (ldf % 4 == 0 && ldw % 4 == 0)),
              "rdm_kpconv_gather: ldw/ldf must be padded");
  if (m == 0) return RDM_OK;
  // form: 0 = the library's choice, 1 = one wavefront per (query, slice) fetching every neighbour row (rounds 1-4), 2 = the
  // support rows of 16 cell-ordered queries staged once in LDS (kpconv_tile_kernel<64, GATHER>, round 5) where it applies
  if (form != 1 && kpconv_tile_gather_applies(c, h, order_records != nullptr) && (form == 2 || tile_gather_pays(c, m)))
    return kpconv_tile_gather(q_points, m, s_points, n_s, s_feats, c, ldf, s_positive, idx, h, ldi, width, kernel_points, sigma, wf,
                              ldw, nn, order_records, i32, stream);
  KpArgs a;
  a.q_points = q_points; a.s_points = s_points; a.s_feats = s_feats; a.s_pos = s_positive;
  a.idx = idx; a.i32 = i32 ? 1 : 0; a.kp = kernel_points; a.width = width; a.wf = wf; a.nn = nn;
  a.order = reinterpret_cast<const float4*>(order_records);
  a.M = static_cast<int>(m); a.Ns = static_cast<int>(n_s); a.H = static_cast<int>(h);
  a.C = static_cast<int>(c); a.ldf = static_cast<int>(ldf); a.ldi = static_cast<int>(ldi);
  a.ldw = static_cast<int>(ldw); a.sigma = sigma; a.split = 1;
  static const bool xcd_env = ::rdm::dev_knob("RDM_GATHER_XCD") != nullptr;  // developer knob (A/B)
  a.xcd_remap = (xcd_env && order_records) ? 1 : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // one work unit per wavefront.  Measured on MI355X: letting a workgroup own 16 or 64 consecutive
  // (cell-ordered) queries for L1 re-use does not help (64: too few wavefronts per CU, 16: neutral)
  auto grid = [&](int split) {
    a.units_per_block = kWaves;
    return dim3(static_cast<unsigned>(ceil_div<int64_t>(m * split, a.units_per_block)));
  };
  // 64 channels (32 for C = 32) per wavefront, four neighbour groups prefetched per trip.  Measured on
  // MI355X: deeper prefetch (24 groups, 3 waves/SIMD) is SLOWER -- the kernel is bound by 128-B line
  // fills from L2 (feature row + point + flag per neighbour), not by the dependent-load chain.
  RDM_DUP_LOOP("gather")
  switch (c) {
    case 1: ::rdm::launch<kpconv_gather_c1_kernel_body, kpconv_gather_c1_kernel, 64 * kWaves>(dim3(ceil_div<int64_t>(m, kWaves)), 0, st, a); break;
    case 32: ::rdm::launch<kpconv_gather_kernel_body<2, 1, 1, 4>, kpconv_gather_kernel<2, 1, 1, 4>, 64 * kWaves>(grid(1), 0, st, a); break;
    case 64: ::rdm::launch<kpconv_gather_kernel_body<4, 1, 1, 4>, kpconv_gather_kernel<4, 1, 1, 4>, 64 * kWaves>(grid(1), 0, st, a); break;
    case 128: ::rdm::launch<kpconv_gather_kernel_body<4, 1, 2, 4>, kpconv_gather_kernel<4, 1, 2, 4>, 64 * kWaves>(grid(2), 0, st, a); break;
    case 256: ::rdm::launch<kpconv_gather_kernel_body<4, 1, 4, 4>, kpconv_gather_kernel<4, 1, 4, 4>, 64 * kWaves>(grid(4), 0, st, a); break;
    case 512: ::rdm::launch<kpconv_gather_kernel_body<4, 2, 4, 2>, kpconv_gather_kernel<4, 2, 4, 2>, 64 * kWaves>(grid(4), 0, st, a); break;
    default:  // any other multiple of 32 (the backbone's widths are 32 * 2^k; the reference takes any init_dim): 32-channel slices
      a.split = static_cast<int>(c / 32);
      ::rdm::launch<kpconv_gather_kernel_body<2, 1, 0, 4>, kpconv_gather_kernel<2, 1, 0, 4>, 64 * kWaves>(grid(a.split), 0, st, a);
      break;
  }
  return launch_status("kpconv_gather_kernel");
}

extern "C" int rdm_kpconv_gather_ordered(const float* q_points, int64_t m, const float* s_points, int64_t n_s,
                                         const float* s_feats, int64_t c, int64_t ldf, const uint8_t* s_positive,
                                         const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                                         const float* kernel_points, float sigma, float* wf, int64_t ldw,
                                         float* nn, const float* order_records, void* stream) {
  return rdm::kpconv_gather_impl(q_points, m, s_points, n_s, s_feats, c, ldf, s_positive, idx, h, ldi, width, kernel_points, sigma,
                                 wf, ldw, nn, order_records, 0, stream);
}

extern "C" int rdm_kpconv_gather_form(const float* q_points, int64_t m, const float* s_points, int64_t n_s,
                                      const float* s_feats, int64_t c, int64_t ldf, const uint8_t* s_positive,
                                      const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                                      const float* kernel_points, float sigma, float* wf, int64_t ldw,
                                      float* nn, const float* order_records, int form, void* stream) {
  RDM_REQUIRE(form >= 0 && form <= 2, "rdm_kpconv_gather_form: form must be 0, 1 or 2");
  return rdm::kpconv_gather_impl(q_points, m, s_points, n_s, s_feats, c, ldf, s_positive, idx, h, ldi, width, kernel_points, sigma,
                                 wf, ldw, nn, order_records, 0, stream, form);
}

extern "C" int rdm_kpconv_gather(const float* q_points, int64_t m, const float* s_points, int64_t n_s,
                                 const float* s_feats, int64_t c, int64_t ldf, const uint8_t* s_positive,
                                 const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                                 const float* kernel_points, float sigma, float* wf, int64_t ldw,
                                 float* nn, void* stream) {
  return rdm_kpconv_gather_ordered(q_points, m, s_points, n_s, s_feats, c, ldf, s_positive, idx, h, ldi, width, kernel_points,
                                   sigma, wf, ldw, nn, nullptr, stream);
}
