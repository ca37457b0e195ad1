// LAB COPY, not built: kpconv_fused.hip with the round-3 producer / consumer variant (kpconv_fused_pc_kernel: one persistent
// workgroup per CU, 12-14 producer wavefronts running a three-stage software pipeline over their queries, 2-4 consumer
// wavefronts contracting parked 16-row blocks; single-writer LDS counters as hand-offs; RDM_PC_TIMING phase stamps read by
// tools/pc_lab.py).  Correct (tests/test_ops_gpu.py -k kpconv passed with it) and measured on MI355X against the lock-step
// kernel that ships (graph-replayed, tools/kpconv_bench.py, us): 32->32 at 32 k queries 65.0 vs 61.3, 32->32 strided 28.4 vs
// 31.0, 64->64 at 11 k 44.6 vs 48.5, 64->64 strided 19.7 vs 18.5-20.5 -- a wash.  The phase stamps say why: under the
// gather's load every memory round trip takes 2.5-5 us, and the W operand (61 / 245 KB re-read from L2 per 16 queries) costs
// the consumers 13 us per block; a later variant with W register-resident measured the same again (docs/EXPERIMENTS.md 5d), so neither the
// phase structure nor W is what bounds these layers.  (A first version with read-modify-write LDS counters
// dead-locked; single-writer counters with release stores / acquire loads are what works.)
// a4 -- KPConv.forward as ONE kernel for the fine levels (C_in = 1, 32, 64): neighbourhood aggregation AND the
// kernel-weight contraction, without the [M, 15*C] intermediate in HBM.
//
// Reference: geotransformer/modules/kpconv/kpconv.py:79-122.  For every query m:
//   WF[m, k, c] = sum_h max(0, 1 - |s[idx[m,h]] - q[m] - kp[k]| / sigma) * feats[idx[m,h], c]      :91-105
//   out[m, c']  = (sum_k sum_c WF[m, k, c] W[k, c, c']) / max(1, #{h : sum_c feats[idx[m,h], c] > 0}) + bias[c']   :107-121
// The two-kernel form (kpconv.hip + gemm.hip) writes WF -- 15*C floats per query, 61 MB for the 32 k-point level at
// C = 32 -- and reads it back.  Here a workgroup aggregates 16 queries (one wavefront per query at a time, the 16x16x4
// MFMA formulation of kpconv.hip), parks the 16 x 15C block in LDS, and multiplies it by W on the same matrix cores:
// the A operand of v_mfma_f32_16x16x4_f32 is a 16-B LDS read (16 queries x 4 consecutive k per instruction), the B
// operand one coalesced 16-B global read per lane from a copy of W stored in operand order
// (rdm_kpconv_pack_weights); the K range is split over the wavefronts and the slices meet in LDS in a fixed order.
// The epilogue divides by the neighbour count, adds the bias, writes the [M, C'] output and accumulates the fp64
// column sums GroupNorm needs (same partial layout as the GEMM epilogue, gemm.hip).
//   C = 32: 16 wavefronts x 1 query, 2 column tiles x 8 K slices; 31 KB block + 32 KB staging -> 2 workgroups per CU
//   C = 64:  8 wavefronts x 2 queries, 4 column tiles x 2 K slices; 62 KB block + 16 KB staging -> 2 workgroups per CU
//   C_in = 1 (first layer, features == 1): no matrix core needed on either side; one wavefront per query, lane = output
//   channel, 16 wavefronts x 4 queries per workgroup.
// Measured (docs/EXPERIMENTS.md §5b, profiles/r02_pmc_fused_kpconv.md): HBM-side writes of these layers drop from 200 MB to 24 MB per
// scan pair, the time does not (the LDS block caps a CU at 16-32 queries in flight in lock-step phases), so the engine
// uses it only when RDM_FUSED_KPCONV=1.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "internal.h"

namespace {

using namespace rdm;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKP = 15;
constexpr int kMaxH = 128;

struct FusedArgs {
  const float* q_points;       // [M,3]
  const float* s_points;       // [Ns,3]
  const float* s_feats;        // [Ns, ldf]
  const unsigned char* s_pos;  // [Ns]
  const int64_t* idx;          // [M, ldi]
  const float* kp;             // [15,3]
  const int32_t* width;        // optional device int: effective row width
  const float* w;              // packed weights (see rdm_kpconv_pack_weights)
  const float* bias;           // [C']
  float* out;                  // [M, ldo]
  double* stats;               // [gridDim.x][2][C'] or null
  int M, Ns, H;
  int ldf, ldi, ldo;
  float sigma;
#ifdef RDM_PC_TIMING
  unsigned long long* clk;  // tools/pc_lab.py: [16 wavefronts][8] shader-clock sums of workgroup 0
#endif
};
#ifdef RDM_PC_TIMING
#define PC_T0() const unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define PC_NOW() __builtin_amdgcn_s_memtime()
#define PC_ADD(slot, t_from) do { const unsigned long long _n = __builtin_amdgcn_s_memtime(); clk_acc[slot] += _n - (t_from); (t_from) = _n; } while (0)
#else
#define PC_ADD(slot, t_from) do { } while (0)
#endif

// ---- C_in in {32, 64}
template <int C, int QB, int NW, int ITERS, int PF>
__global__ __launch_bounds__(64 * NW) void kpconv_fused_kernel(FusedArgs a) {
  constexpr int VEC = C / 16;           // channels per lane and gather tile pass (channel = VEC*j + e)
  constexpr int NT = C / 16;            // output column tiles (C' = C)
  constexpr int RT = QB / 16;           // output row tiles
  constexpr int TILES = NT * RT;        // output tiles of 16 x 16
  constexpr int KS = NW / TILES;        // K split over wavefronts
  constexpr int K16 = kKP * C / 16;     // 16-deep contraction steps (30 / 60)
  constexpr int LDW = kKP * C + 4;      // LDS row stride of the aggregated block (floats)
  constexpr int QPW = QB / NW;          // queries per wavefront and iteration
  // PF = neighbour groups (of four) fetched per trip.  LDS, not registers, sets the residency of this kernel, so the
  // C = 64 instance spends registers on eight groups in flight (the stand-alone gather keeps four: kpconv.hip)
  constexpr int STEPS = (K16 + KS - 1) / KS;  // contraction steps of a wavefront (at most)
  constexpr int WD = STEPS < 6 ? STEPS : 6;   // W operand loads a wavefront keeps in flight
  static_assert(TILES * KS == NW && QPW * NW == QB && (KS - 1) * TILES * 1024 <= NW * kMaxH * 16, "wavefront roles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* WF = smem;                                                        // [QB][LDW]
  float4* nb_all = reinterpret_cast<float4*>(smem + QB * LDW);             // [NW][kMaxH]: rel.xyz, w = support row
  float* nn_s = smem + QB * LDW + NW * kMaxH * 4;                          // [QB]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  float4* nb = nb_all + wave * kMaxH;
  const float kx = j < kKP ? a.kp[3 * j] : 0.f, ky = j < kKP ? a.kp[3 * j + 1] : 0.f, kz = j < kKP ? a.kp[3 * j + 2] : 0.f;
  const float inv_sigma = 1.0f / a.sigma;  // (hardware sqrt + reciprocal multiply as in kpconv.hip)
  int H = a.H;
  if (a.width) H = min(H, *a.width);
  // roles of the contraction phase
  const int tw = wave % TILES, kh = wave / TILES, rt = tw / NT, ct = tw % NT;
  const float bias_v = a.bias[16 * ct + j];
  double st_s = 0.0, st_ss = 0.0;  // GroupNorm column sums of this lane's column (rows 4g .. 4g+3 of every row tile it owns)

  for (int it = 0; it < ITERS; ++it) {
    const int q0 = (blockIdx.x * ITERS + it) * QB;
    if (q0 >= a.M) break;  // (uniform over the workgroup)
    // ------------------------------------------------------------ aggregation: QPW queries per wavefront
    for (int qq = 0; qq < QPW; ++qq) {
      const int ql = wave * QPW + qq, m = q0 + ql;
      if (m >= a.M) break;
      const float qx = a.q_points[3 * m], qy = a.q_points[3 * m + 1], qz = a.q_points[3 * m + 2];
      int positives = 0;
      f32x4 acc[VEC];
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int hc = 0; hc < H; hc += kMaxH) {  // chunks of the staging row (one for every KITTI limit), as in kpconv.hip
        const int Hc = min(H - hc, kMaxH);
        int Hq = 0;  // slots up to the last real neighbour (kpconv.hip: shadow slots contribute exact zeros)
        for (int hb = 0; hb < Hc; hb += 64) {
          const int h = hb + lane;
          const int64_t id = h < Hc ? a.idx[static_cast<int64_t>(m) * a.ldi + hc + h] : -1;
          const unsigned long long rm = __builtin_amdgcn_ballot_w64(id >= 0 && id < a.Ns);
          if (rm) Hq = hb + 64 - __builtin_clzll(rm);
          float4 v;
          if (id >= 0 && id < a.Ns) {
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
          if (h < Hc) nb[h] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int h0 = 0; h0 < Hq; h0 += 4 * PF) {
          float w[PF];
          float f[PF][VEC];
#pragma unroll
          for (int p = 0; p < PF; ++p) {
            const int h = h0 + 4 * p + g;
            int id = -1;
            w[p] = 0.f;
            if (h < Hq) {
              const float4 v = nb[h];
              id = __float_as_int(v.w);
              const float dx = v.x - kx, dy = v.y - ky, dz = v.z - kz;
              const float d2 = (dx * dx + dy * dy) + dz * dz;
              w[p] = fmaxf(0.f, 1.f - __builtin_amdgcn_sqrtf(d2) * inv_sigma);
              if (j >= kKP || id < 0) w[p] = 0.f;
            }
            const float* row = a.s_feats + static_cast<int64_t>(id < 0 ? 0 : id) * a.ldf + VEC * j;
            if (id >= 0) {
              if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(row);
                f[p][0] = t.x; f[p][1] = t.y; f[p][2] = t.z; f[p][3] = t.w;
              } else {
                const float2 t = *reinterpret_cast<const float2*>(row);
                f[p][0] = t.x; f[p][1] = t.y;
              }
            } else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) f[p][e] = 0.f;
            }
          }
#pragma unroll
          for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p], f[p][e], acc[e], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // nb is rewritten by the next chunk / query
        __builtin_amdgcn_wave_barrier();
      }
      positives = wave_sum_i(positives);
      // park WF[ql, k, c]: accumulator row 4g + r = kernel point, lane j holds channels VEC*j .. VEC*j + VEC-1
      float* dst = WF + ql * LDW + VEC * j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * g + r;
        if (k >= kKP) continue;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst + k * C) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        else *reinterpret_cast<float2*>(dst + k * C) = make_float2(acc[0][r], acc[1][r]);
      }
      if (lane == 0) nn_s[ql] = static_cast<float>(positives > 1 ? positives : 1);
    }
    __syncthreads();

    // ------------------------------------------------------------ contraction: out[QB, C'] = WF[QB, 15C] W[15C, C']
    // wavefront (rt, ct, kh): row tile rt, column tile ct, K steps [kh, kh+1) * K16 / KS.  Step s covers k = 16 s ..
    // 16 s + 15 in the order lane group g -> k = 16 s + 4 g + e (e = the e-th of four MFMAs): A is one 16-B LDS read,
    // B one 16-B global read of the packed weights.
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    {
      const float* arow = WF + (16 * rt + j) * LDW + 4 * g;
      const float4* wp = reinterpret_cast<const float4*>(a.w) + static_cast<int64_t>(ct) * 64 + lane;
      const int s_begin = kh * K16 / KS, s_end = (kh + 1) * K16 / KS;
      // a ring of WD operand loads in flight (a fully unrolled loop had kept all 30 of the C = 64 instance in registers:
      // 186 VGPRs, one workgroup per CU; the ring needs 100 and two workgroups share a CU)
      if constexpr (STEPS <= 4) {  // (C = 32: three or four steps per wavefront, all loads issued up front)
#pragma unroll 4
        for (int s = s_begin; s < s_end; ++s) {
          const float4 bv = wp[static_cast<int64_t>(s) * NT * 64];
          const float4 av = *reinterpret_cast<const float4*>(arow + 16 * s);
          o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, o1, 0, 0, 0);
          o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, o1, 0, 0, 0);
        }
      } else {
      float4 bq[WD];
#pragma unroll
      for (int i = 0; i < WD; ++i) bq[i] = wp[static_cast<int64_t>(min(s_begin + i, s_end - 1)) * NT * 64];
#pragma unroll 1
      for (int s0 = s_begin; s0 < s_end; s0 += WD) {
#pragma unroll
        for (int i = 0; i < WD; ++i) {
          const int s = s0 + i;
          if (s < s_end) {  // (wavefront-uniform)
            const float4 bv = bq[i];
            const float4 av = *reinterpret_cast<const float4*>(arow + 16 * s);
            bq[i] = wp[static_cast<int64_t>(min(s + WD, s_end - 1)) * NT * 64];
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, o1, 0, 0, 0);
          }
        }
      }
      }
    }
    f32x4 o = o0 + o1;
    // the K slices of a tile meet in the neighbour-staging area (idle since the barrier above; the barrier that ends the
    // iteration keeps the next aggregation from overwriting it early) and are added in slice order
    f32x4* red = reinterpret_cast<f32x4*>(nb_all);  // [(KS - 1) * TILES][64]
    if (kh > 0) red[((kh - 1) * TILES + tw) * 64 + lane] = o;
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int k = 1; k < KS; ++k) o = o + red[((k - 1) * TILES + tw) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * rt + 4 * g + r, m = q0 + ql;
        if (m < a.M) {
          float v = o[r] / nn_s[ql];
          v += bias_v;
          a.out[static_cast<int64_t>(m) * a.ldo + 16 * ct + j] = v;
          st_s += static_cast<double>(v);
          st_ss += static_cast<double>(v) * static_cast<double>(v);
        }
      }
    }
    __syncthreads();  // WF and nn_s are rewritten by the next iteration
  }
  if (a.stats) {  // column sums of this workgroup's rows: lane groups g, then row tiles, in a fixed order
    st_s = (st_s + __shfl_xor(st_s, 16, 64)) + (__shfl_xor(st_s, 32, 64) + __shfl_xor(st_s, 48, 64));
    st_ss = (st_ss + __shfl_xor(st_ss, 16, 64)) + (__shfl_xor(st_ss, 32, 64) + __shfl_xor(st_ss, 48, 64));
    double* ex = reinterpret_cast<double*>(smem);  // [TILES][16][2]
    if (kh == 0 && g == 0) {
      ex[(tw * 16 + j) * 2 + 0] = st_s;
      ex[(tw * 16 + j) * 2 + 1] = st_ss;
    }
    __syncthreads();
    if (threadIdx.x < C) {
      const int c = threadIdx.x, cti = c / 16, cj = c % 16;
      double s = 0.0, ss = 0.0;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        s += ex[((r * NT + cti) * 16 + cj) * 2 + 0];
        ss += ex[((r * NT + cti) * 16 + cj) * 2 + 1];
      }
      a.stats[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * C + c] = s;
      a.stats[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * C + c] = ss;
    }
  }
}

// ---- C_in in {32, 64}, round 3: gather and contraction OVERLAPPED inside a CU (producer / consumer wavefronts)
// The kernel above runs its workgroup in lock step -- aggregate 16 queries, barrier, contract, barrier -- and measured as the
// SUM of its phases (48 us for the 11 k-point 64 -> 64 layer against 26 us for the stand-alone gather, tools/kpconv_bench.py):
// while a workgroup contracts, its wavefronts gather nothing.  Here ONE persistent workgroup per CU splits its 16 wavefronts
// into roles: NP producers, dealt the queries of the workgroup's blocks round-robin and running the
// aggregation of kpconv.hip (PF neighbour groups in flight) without ever meeting a workgroup barrier, and C/16 consumers,
// one per 16-column output tile, which wait for a block of 16 parked rows, multiply it by W (operand loads in a ring of six)
// and write the output rows and the GroupNorm partial sums.  Blocks travel through a ring of S LDS slots; the hand-offs are
// single-writer LDS counters that only grow (queries parked per producer, blocks drained per consumer), release stores and
// acquire loads at workgroup scope, s_sleep while waiting.  Results do not depend on the schedule: every output row is a fixed sequence of
// operations on its own data, and a consumer's partial sums run over its blocks in order.
constexpr int kStage = 128;  // neighbour slots staged per producer (= kMaxH: wider rows run in chunks)

template <int C, int NP, int S, int PF, int WD>
__global__ __launch_bounds__(1024) void kpconv_fused_pc_kernel(FusedArgs a) {
  constexpr int VEC = C / 16;           // channels per lane and gather tile pass (channel = VEC*j + e)
  constexpr int NC = C / 16;            // consumers = output column tiles (C' = C)
  constexpr int K16 = kKP * C / 16;     // 16-deep contraction steps (30 / 60)
  constexpr int LDW = kKP * C + 4;      // LDS row stride of a parked row (floats)
  // WD = W operand loads a consumer keeps in flight: a block needs K16 of them (60 KB for C = 64) and a ring of six left the
  // consumers latency-bound (10 round trips per block: they, not the producers, set the kernel's time)
  static_assert(NP + NC == 16 && K16 % WD == 0, "wavefront roles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* WF = smem;                                                        // [S][16][LDW]
  float4* nb_all = reinterpret_cast<float4*>(smem + S * 16 * LDW);         // [NP][kStage]: rel.xyz, w = support row
  float* nn_s = smem + S * 16 * LDW + NP * kStage * 4;                     // [S][16]
  // hand-off counters, one writer each (plain release stores, no read-modify-write): parked[p] = queries producer p has
  // parked, drained[c] = blocks consumer c has finished
  int* parked = reinterpret_cast<int*>(nn_s + S * 16);                     // [NP]
  int* drained = parked + NP;                                              // [NC]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // (wave: a scalar -> scalar branches)
  const int g = lane >> 4, j = lane & 15;
  if (threadIdx.x < 16) parked[threadIdx.x] = 0;
  __syncthreads();  // (the only workgroup barrier)
  const int nblocks = (a.M + 15) / 16;
  const int G = gridDim.x;
  const int nb_wg = (nblocks - static_cast<int>(blockIdx.x) + G - 1) / G;  // blocks blockIdx.x, blockIdx.x + G, ...
  int H = a.H;
  if (a.width) H = min(H, *a.width);

  if (wave < NP) {
    // ------------------------------------------------------------------ producer: a three-stage pipeline over its queries
    // A query costs five dependent memory round trips when handled alone -- its index row, its neighbours' points and
    // flags, then the feature rows in trips of PF groups -- and it is those round trips, not bandwidth, that bound the
    // aggregation (a wavefront of the stand-alone gather spends ~14 us per query).  Here the index row of query k + 2 and
    // the points of query k + 1 are in flight while query k's feature rows are fetched and multiplied: stage A (indices),
    // stage B (points, flags) and stage C (LDS staging, influences, MFMA) of three consecutive queries overlap, in
    // registers (two buffers per stage, the loop is unrolled twice so that every buffer has a compile-time name).
    // The pipeline covers the first 128 slots of a row (every calibrated KITTI limit); later chunks of wider rows are fetched
    // inside stage C.
    float4* nb = nb_all + wave * kStage;
    const float kx = j < kKP ? a.kp[3 * j] : 0.f, ky = j < kKP ? a.kp[3 * j + 1] : 0.f, kz = j < kKP ? a.kp[3 * j + 2] : 0.f;
    const float inv_sigma = 1.0f / a.sigma;
    struct QA { long long id0, id1; int m; };
    struct QB { float px0, py0, pz0, px1, py1, pz1; int pos0, pos1, id0, id1, m; };
    const int t_end = 16 * nb_wg;
    auto query_row = [&](int k) {  // global row of this producer's k-th query, -1 past the end
      const int t = wave + k * NP;
      if (t >= t_end) return -1;
      const int m = (static_cast<int>(blockIdx.x) + (t >> 4) * G) * 16 + (t & 15);
      return m < a.M ? m : -1;
    };
    auto issue_a = [&](int k, QA& A) __attribute__((always_inline)) {
      A.m = query_row(k);
      A.id0 = A.id1 = -1;
      if (A.m >= 0) {
        const int64_t* row = a.idx + static_cast<int64_t>(A.m) * a.ldi;
        if (lane < H) A.id0 = row[lane];
        if (lane + 64 < H) A.id1 = row[lane + 64];
      }
    };
    auto issue_b = [&](const QA& A, QB& B) __attribute__((always_inline)) {
      B.m = A.m;
      B.id0 = (A.id0 >= 0 && A.id0 < a.Ns) ? static_cast<int>(A.id0) : -1;
      B.id1 = (A.id1 >= 0 && A.id1 < a.Ns) ? static_cast<int>(A.id1) : -1;
      B.px0 = B.py0 = B.pz0 = B.px1 = B.py1 = B.pz1 = 1.0e6f;  // shadow neighbour: point at 1e6, zero features
      B.pos0 = B.pos1 = 0;
      if (B.id0 >= 0) {
        B.px0 = a.s_points[3 * B.id0]; B.py0 = a.s_points[3 * B.id0 + 1]; B.pz0 = a.s_points[3 * B.id0 + 2];
        B.pos0 = a.s_pos[B.id0];
      }
      if (B.id1 >= 0) {
        B.px1 = a.s_points[3 * B.id1]; B.py1 = a.s_points[3 * B.id1 + 1]; B.pz1 = a.s_points[3 * B.id1 + 2];
        B.pos1 = a.s_pos[B.id1];
      }
    };
    int n_parked = 0;
#ifdef RDM_PC_TIMING
    unsigned long long clk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tcur = PC_NOW();
    const unsigned long long tstart = tcur;
#endif
    auto stage_c = [&](int k, const QB& B) __attribute__((always_inline)) {
      PC_ADD(1, tcur);  // issue of stages A / B
      const int t = wave + k * NP;
      const int seq = t >> 4, ql = t & 15, slot = seq % S;
      const int m = B.m;
      f32x4 acc[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      int positives = 0;
      if (m >= 0) {
        const float qx = a.q_points[3 * m], qy = a.q_points[3 * m + 1], qz = a.q_points[3 * m + 2];
        int pos_lane = B.pos0 + B.pos1;
        for (int hc = 0; hc < H; hc += kStage) {  // one chunk for every KITTI limit; later chunks of wider rows are not pipelined
        int Hq;  // slots of the chunk up to its last real neighbour (shadow slots contribute exact zeros)
        if (hc == 0) {
          const unsigned long long r0 = __builtin_amdgcn_ballot_w64(B.id0 >= 0), r1 = __builtin_amdgcn_ballot_w64(B.id1 >= 0);
          Hq = r1 ? 128 - __builtin_clzll(r1) : (r0 ? 64 - __builtin_clzll(r0) : 0);
          if (lane < H) nb[lane] = make_float4(B.px0 - qx, B.py0 - qy, B.pz0 - qz, __int_as_float(B.id0));
          if (lane + 64 < H) nb[lane + 64] = make_float4(B.px1 - qx, B.py1 - qy, B.pz1 - qz, __int_as_float(B.id1));
        } else {
          Hq = 0;
          const int Hc = min(H - hc, kStage);
          for (int hb = 0; hb < Hc; hb += 64) {
            const int h = hb + lane;
            const int64_t id = h < Hc ? a.idx[static_cast<int64_t>(m) * a.ldi + hc + h] : -1;
            const bool real = id >= 0 && id < a.Ns;
            const unsigned long long rm = __builtin_amdgcn_ballot_w64(real);
            if (rm) Hq = hb + 64 - __builtin_clzll(rm);
            float4 v = make_float4(1.0e6f - qx, 1.0e6f - qy, 1.0e6f - qz, __int_as_float(-1));
            if (real) {
              v = make_float4(a.s_points[3 * id] - qx, a.s_points[3 * id + 1] - qy, a.s_points[3 * id + 2] - qz,
                              __int_as_float(static_cast<int>(id)));
              pos_lane += a.s_pos[id];
            }
            if (h < Hc) nb[h] = v;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        PC_ADD(2, tcur);  // wait for stage B's data, LDS staging
        for (int h0 = 0; h0 < Hq; h0 += 4 * PF) {
          float w[PF];
          float f[PF][VEC];
#pragma unroll
          for (int p = 0; p < PF; ++p) {
            const int h = h0 + 4 * p + g;
            int id = -1;
            w[p] = 0.f;
            if (h < Hq) {
              const float4 v = nb[h];
              id = __float_as_int(v.w);
              const float dx = v.x - kx, dy = v.y - ky, dz = v.z - kz;
              const float d2 = (dx * dx + dy * dy) + dz * dz;
              w[p] = fmaxf(0.f, 1.f - __builtin_amdgcn_sqrtf(d2) * inv_sigma);
              if (j >= kKP || id < 0) w[p] = 0.f;
            }
            const float* row = a.s_feats + static_cast<int64_t>(id < 0 ? 0 : id) * a.ldf + VEC * j;
            if (id >= 0) {
              if constexpr (VEC == 4) {
                const float4 tt = *reinterpret_cast<const float4*>(row);
                f[p][0] = tt.x; f[p][1] = tt.y; f[p][2] = tt.z; f[p][3] = tt.w;
              } else {
                const float2 tt = *reinterpret_cast<const float2*>(row);
                f[p][0] = tt.x; f[p][1] = tt.y;
              }
            } else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) f[p][e] = 0.f;
            }
          }
#pragma unroll
          for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p], f[p][e], acc[e], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // nb is rewritten by the next chunk / query
        __builtin_amdgcn_wave_barrier();
        }  // chunk loop
        positives = wave_sum_i(pos_lane);
      }
      PC_ADD(3, tcur);  // feature-row trips + MFMA
      // the slot of this block is free once every consumer has drained its previous tenant (block seq - S)
      if (seq >= S) {  // lanes 0 .. NC-1 watch one consumer each
        const int need = seq - S + 1;
        while (__builtin_amdgcn_ballot_w64(lane < NC && __hip_atomic_load(drained + (lane < NC ? lane : 0), __ATOMIC_ACQUIRE,
                                                                          __HIP_MEMORY_SCOPE_WORKGROUP) < need))
          __builtin_amdgcn_s_sleep(2);
      }
      PC_ADD(4, tcur);  // wait for the slot
      if (m >= 0) {  // park WF[ql, k, c]: accumulator row 4g + r = kernel point, lane j holds channels VEC*j .. VEC*j + VEC-1
        float* dst = WF + (slot * 16 + ql) * LDW + VEC * j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = 4 * g + r;
          if (kk >= kKP) continue;
          if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst + kk * C) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
          else *reinterpret_cast<float2*>(dst + kk * C) = make_float2(acc[0][r], acc[1][r]);
        }
        if (lane == 0) nn_s[slot * 16 + ql] = static_cast<float>(positives > 1 ? positives : 1);
      }
      // (rows past the end of the tensor count too)
      ++n_parked;
      if (lane == 0) __hip_atomic_store(parked + wave, n_parked, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      PC_ADD(5, tcur);  // park + publish
    };
    const int nq = (t_end - wave + NP - 1) / NP;  // queries of this producer
    QA a0, a1;
    QB b0, b1;
    issue_a(0, a0);
    issue_a(1, a1);
    issue_b(a0, b0);
    for (int k = 0; k < nq; k += 2) {
      issue_b(a1, b1);      // points of query k + 1
      issue_a(k + 2, a0);   // index row of query k + 2
      stage_c(k, b0);
      if (k + 1 >= nq) break;
      issue_b(a0, b0);      // points of query k + 2
      issue_a(k + 3, a1);   // index row of query k + 3
      stage_c(k + 1, b1);
    }
#ifdef RDM_PC_TIMING
    if (a.clk && blockIdx.x == 0 && lane == 0) {
      clk_acc[0] = PC_NOW() - tstart;
      clk_acc[6] = nq;
      for (int i = 0; i < 8; ++i) a.clk[wave * 8 + i] = clk_acc[i];
    }
#endif
    return;
  }

  // -------------------------------------------------------------------- consumer: output columns [16 ct, 16 ct + 16)
  const int ct = wave - NP;
  const float bias_v = a.bias[16 * ct + j];
  const float4* wp = reinterpret_cast<const float4*>(a.w) + static_cast<int64_t>(ct) * 64 + lane;
  double st_s = 0.0, st_ss = 0.0;  // GroupNorm column sums of this lane's column over the rows 4g .. 4g+3 of its blocks
#ifdef RDM_PC_TIMING
  unsigned long long clk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tcur = PC_NOW();
  const unsigned long long tstart = tcur;
#endif
  for (int seq = 0; seq < nb_wg; ++seq) {
    const int slot = seq % S;
    const int q0 = (static_cast<int>(blockIdx.x) + seq * G) * 16;
    // the block's first W operands are requested before the wait: they do not depend on the producers
    float4 bq[WD];
#pragma unroll
    for (int i = 0; i < WD; ++i) bq[i] = wp[static_cast<int64_t>(i) * NC * 64];
    {  // all 16 rows of the block parked: lane p < NP watches producer p, which owns the queries t = p (mod NP)
      const int t_end = 16 * (seq + 1);  // producer p has parked ceil((t_end - p) / NP) queries by then
      const int need = lane < NP ? (t_end - lane + NP - 1) / NP : 0;
      while (__builtin_amdgcn_ballot_w64(lane < NP && __hip_atomic_load(parked + (lane < NP ? lane : 0), __ATOMIC_ACQUIRE,
                                                                        __HIP_MEMORY_SCOPE_WORKGROUP) < need))
        __builtin_amdgcn_s_sleep(2);
    }
    PC_ADD(1, tcur);  // wait for the block
    // out[16, 16] = WF[16, 15C] W[15C, 16 ct ..]: step s covers k = 16 s .. 16 s + 15 in the order lane group g -> k = 16 s +
    // 4 g + e (e = the e-th of four MFMAs): A is one 16-B LDS read, B one 16-B global read of the packed weights
    const float* arow = WF + (slot * 16 + j) * LDW + 4 * g;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int s0 = 0; s0 < K16; s0 += WD) {
#pragma unroll
      for (int i = 0; i < WD; ++i) {
        const int s = s0 + i;
        const float4 bv = bq[i];
        const float4 av = *reinterpret_cast<const float4*>(arow + 16 * s);
        bq[i] = wp[static_cast<int64_t>(min(s + WD, K16 - 1)) * NC * 64];
        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, o1, 0, 0, 0);
      }
    }
    const f32x4 o = o0 + o1;
    float nn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) nn[r] = nn_s[slot * 16 + 4 * g + r];
    // this wavefront is done with the slot (its reads of WF and nn_s have returned: the values are used below)
    if (lane == 0) __hip_atomic_store(drained + ct, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    PC_ADD(2, tcur);  // contraction
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = q0 + 4 * g + r;
      if (m < a.M) {
        float v = o[r] / nn[r];
        v += bias_v;
        a.out[static_cast<int64_t>(m) * a.ldo + 16 * ct + j] = v;
        st_s += static_cast<double>(v);
        st_ss += static_cast<double>(v) * static_cast<double>(v);
      }
    }
    PC_ADD(3, tcur);  // epilogue
  }
#ifdef RDM_PC_TIMING
  if (a.clk && blockIdx.x == 0 && lane == 0) {
    clk_acc[0] = PC_NOW() - tstart;
    clk_acc[6] = nb_wg;
    for (int i = 0; i < 8; ++i) a.clk[wave * 8 + i] = clk_acc[i];
  }
#endif
  if (a.stats) {  // column sums of this workgroup's rows: lane groups g in a fixed order
    st_s = (st_s + __shfl_xor(st_s, 16, 64)) + (__shfl_xor(st_s, 32, 64) + __shfl_xor(st_s, 48, 64));
    st_ss = (st_ss + __shfl_xor(st_ss, 16, 64)) + (__shfl_xor(st_ss, 32, 64) + __shfl_xor(st_ss, 48, 64));
    if (g == 0) {
      a.stats[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * C + 16 * ct + j] = st_s;
      a.stats[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * C + 16 * ct + j] = st_ss;
    }
  }
}

template <int C, int NP, int S>
constexpr size_t fused_pc_lds_bytes() {
  return sizeof(float) * (static_cast<size_t>(S) * 16 * (kKP * C + 4) + NP * kStage * 4 + S * 16) + sizeof(int) * 16 + 16;
}
// C = 32: 14 producers + 2 consumers, four slots of 31 KB; C = 64: 12 producers + 4 consumers, two slots of 62 KB (150 KB of LDS
// either way: one workgroup = 16 wavefronts per CU, 128 VGPRs each)
constexpr int kPcNp32 = 14, kPcS32 = 4, kPcPf32 = 8, kPcWd32 = 15, kPcNp64 = 12, kPcS64 = 2, kPcPf64 = 8, kPcWd64 = 12;

// ---- C_in = 1: one wavefront per query, lane = output channel (C' = 64); QPW queries per wavefront
constexpr int kC1Out = 64, kC1Waves = 16, kC1Qpw = 4;  // 64 queries per workgroup, four per wavefront
__global__ __launch_bounds__(64 * kC1Waves) void kpconv_fused_c1_kernel(FusedArgs a) {
  __shared__ float4 nb_all[kC1Waves][kMaxH];  // rel.xyz, w = feature (0 for shadow neighbours)
  __shared__ double ex[kC1Waves][kC1Out][2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  float4* nb = nb_all[wave];
  const float kx = j < kKP ? a.kp[3 * j] : 0.f, ky = j < kKP ? a.kp[3 * j + 1] : 0.f, kz = j < kKP ? a.kp[3 * j + 2] : 0.f;
  const float inv_sigma = 1.0f / a.sigma;
  int H = a.H;
  if (a.width) H = min(H, *a.width);
  float wcol[kKP];  // W[k][lane]
#pragma unroll
  for (int k = 0; k < kKP; ++k) wcol[k] = a.w[k * kC1Out + lane];
  const float bias_v = a.bias[lane];
  double st_s = 0.0, st_ss = 0.0;
  const int m0 = (blockIdx.x * kC1Waves + wave) * kC1Qpw;
  for (int qq = 0; qq < kC1Qpw; ++qq) {
    const int m = m0 + qq;
    if (m >= a.M) break;
    const float qx = a.q_points[3 * m], qy = a.q_points[3 * m + 1], qz = a.q_points[3 * m + 2];
    int positives = 0;
    float acc = 0.f;  // lane (g, j): kernel point j over neighbours g, g+4, ...
    for (int hc = 0; hc < H; hc += kMaxH) {  // chunks of the staging row
      const int Hc = min(H - hc, kMaxH);
      int Hq = 0;  // slots up to the last real neighbour
      for (int hb = 0; hb < Hc; hb += 64) {
        const int h = hb + lane;
        const int64_t id = h < Hc ? a.idx[static_cast<int64_t>(m) * a.ldi + hc + h] : -1;
        const unsigned long long rm = __builtin_amdgcn_ballot_w64(id >= 0 && id < a.Ns);
        if (rm) Hq = hb + 64 - __builtin_clzll(rm);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0 && id < a.Ns) {
          v.x = a.s_points[3 * id] - qx;
          v.y = a.s_points[3 * id + 1] - qy;
          v.z = a.s_points[3 * id + 2] - qz;
          v.w = a.s_feats[id * a.ldf];
          positives += a.s_pos[id];
        }
        if (h < Hc) nb[h] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int h = g; h < Hq; h += 4) {
        const float4 v = nb[h];
        const float dx = v.x - kx, dy = v.y - ky, dz = v.z - kz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        acc += fmaxf(0.f, 1.f - __builtin_amdgcn_sqrtf(d2) * inv_sigma) * v.w;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // nb is rewritten by the next chunk / query
      __builtin_amdgcn_wave_barrier();
    }
    positives = wave_sum_i(positives);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);  // every lane (., j) now holds WF[k = j]
    float o = 0.f;
#pragma unroll
    for (int k = 0; k < kKP; ++k)
      o = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), k)), wcol[k], o);
    float v = o / static_cast<float>(positives > 1 ? positives : 1);
    v += bias_v;
    a.out[static_cast<int64_t>(m) * a.ldo + lane] = v;
    st_s += static_cast<double>(v);
    st_ss += static_cast<double>(v) * static_cast<double>(v);
  }
  if (a.stats) {
    ex[wave][lane][0] = st_s;
    ex[wave][lane][1] = st_ss;
    __syncthreads();
    if (wave == 0) {
      double s = 0.0, ss = 0.0;
#pragma unroll
      for (int w = 0; w < kC1Waves; ++w) {  // fixed order
        s += ex[w][lane][0];
        ss += ex[w][lane][1];
      }
      a.stats[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * kC1Out + lane] = s;
      a.stats[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * kC1Out + lane] = ss;
    }
  }
}

template <int C, int QB, int NW>
constexpr size_t fused_lds_bytes() { return sizeof(float) * (static_cast<size_t>(QB) * (kKP * C + 4) + NW * kMaxH * 4 + QB); }

// C = 32: 16 wavefronts x 1 query, 31 KB block + 32 KB staging = 63 KB -> two workgroups (32 wavefronts) per CU, the
// occupancy of the stand-alone gather; C = 64: the 62 KB block leaves room for 8 wavefronts x 2 queries (78 KB, two per CU)
#ifndef RDM_F64_NW  // (tools/ab_fused_variants.sh builds other shapes of the C = 64 instance)
#define RDM_F64_NW 8
#define RDM_F64_PF 8
#endif
constexpr int kQb32 = 16, kNw32 = 16, kIters32 = 2, kPf32 = 4, kQb64 = 16, kNw64 = RDM_F64_NW, kIters64 = 1, kPf64 = RDM_F64_PF;

}  // namespace

// Whether the engine and the per-op path route the fine levels (C_in = 1, 32, 64) through the fused kernel.  ON by
// default since round 3: the [M, 15*C] tensor between gather and weight product was a 13x write amplification of
// these layers (profiles/r02_pmc_fetch_write.md); RDM_FUSED_KPCONV=0 selects the two-kernel form for A/B runs.
extern "C" int rdm_kpconv_fused_enabled(void) {
  static const bool on = [] {
    const char* v = ::rdm::dev_knob("RDM_FUSED_KPCONV");
    return !(v != nullptr && v[0] == '0');
  }();
  return on ? 1 : 0;
}

extern "C" int rdm_kpconv_fused_supported(int64_t c_in, int64_t c_out) {
  return (c_in == 1 && c_out == kC1Out) || (c_in == 32 && c_out == 32) || (c_in == 64 && c_out == 64);
}

#ifdef RDM_PC_TIMING
static unsigned long long* g_pc_clk = nullptr;
extern "C" void rdm_dbg_pc_timing(unsigned long long* device_buf) { g_pc_clk = device_buf; }
#endif
namespace {
int cu_count() {  // CUs of the current device (one persistent workgroup each in the producer / consumer kernels)
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return cus;
  }();
  return n;
}
bool lockstep_form() {  // lab build: RDM_FUSED_LOCKSTEP=1 selects the round-2 kernel (barrier-separated phases) for A/B runs
  static const bool on = [] { const char* v = ::rdm::dev_knob("RDM_FUSED_LOCKSTEP"); return v && v[0] == '1'; }();
  return on;
}
int64_t rows_per_block_lockstep(int64_t c_in) {
  return c_in == 1 ? kC1Waves * kC1Qpw : (c_in == 32 ? kQb32 * kIters32 : kQb64 * kIters64);
}
}  // namespace

// Rows of the fp64 GroupNorm partial array [rows][2][c_out] a call with m queries writes (one per workgroup).
extern "C" int64_t rdm_kpconv_fused_partial_rows(int64_t m, int64_t c_in) {
  if (m <= 0) return 0;
  if (c_in == 1 || lockstep_form()) return rdm::ceil_div<int64_t>(m, rows_per_block_lockstep(c_in));
  return std::min<int64_t>(rdm::ceil_div<int64_t>(m, 16), cu_count());
}

extern "C" size_t rdm_kpconv_packed_floats(int64_t c_in, int64_t c_out) {
  return c_in == 1 ? static_cast<size_t>(16) * c_out : static_cast<size_t>(kKP) * c_in * c_out;
}

// w [15, c_in, c_out] (the checkpoint layout, host) -> the B-operand order of the fused kernel (host):
//   c_in = 1: [16, c_out] (row 15 zero);  otherwise float4 records [s][column tile][lane = 16 g + j] holding
//   W[16 s + 4 g + e][16 tile + j], e = 0..3 (the row index of W is k * c_in + c).
extern "C" int rdm_kpconv_pack_weights(const float* w, int64_t c_in, int64_t c_out, float* packed) {
  using namespace rdm;
  RDM_REQUIRE(w && packed && rdm_kpconv_fused_supported(c_in, c_out), "rdm_kpconv_pack_weights: unsupported (%lld -> %lld)",
              (long long)c_in, (long long)c_out);
  if (c_in == 1) {
    for (int64_t k = 0; k < 16; ++k)
      for (int64_t c = 0; c < c_out; ++c) packed[k * c_out + c] = k < kKP ? w[k * c_out + c] : 0.f;
    return RDM_OK;
  }
  const int64_t nt = c_out / 16, k16 = kKP * c_in / 16;
  for (int64_t s = 0; s < k16; ++s)
    for (int64_t t = 0; t < nt; ++t)
      for (int64_t lane = 0; lane < 64; ++lane)
        for (int64_t e = 0; e < 4; ++e)
          packed[((s * nt + t) * 64 + lane) * 4 + e] = w[(16 * s + 4 * (lane >> 4) + e) * c_out + 16 * t + (lane & 15)];
  return RDM_OK;
}

extern "C" int rdm_kpconv_fused(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats,
                                int64_t c, int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h,
                                int64_t ldi, const int32_t* width, const float* kernel_points, float sigma,
                                const float* w_packed, const float* bias, int64_t c_out, float* out, int64_t ldo,
                                double* gn_partial, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(q_points && s_points && s_feats && s_positive && idx && kernel_points && w_packed && bias && out,
              "rdm_kpconv_fused: null pointer");
  RDM_REQUIRE(rdm_kpconv_fused_supported(c, c_out), "rdm_kpconv_fused: unsupported channel counts %lld -> %lld", (long long)c,
              (long long)c_out);
  RDM_REQUIRE(m >= 0 && n_s > 0 && h > 0 && ldo >= c_out, "rdm_kpconv_fused: bad sizes (h=%lld)", (long long)h);
  RDM_REQUIRE(c == 1 || (ldf % 4 == 0 && (reinterpret_cast<uintptr_t>(s_feats) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(w_packed) & 15) == 0),
              "rdm_kpconv_fused: features / packed weights must be 16-byte aligned with a row stride that is a multiple of 4");
  if (m == 0) return RDM_OK;
  FusedArgs a;
  a.q_points = q_points; a.s_points = s_points; a.s_feats = s_feats; a.s_pos = s_positive; a.idx = idx; a.kp = kernel_points;
  a.width = width; a.w = w_packed; a.bias = bias; a.out = out; a.stats = gn_partial;
  a.M = static_cast<int>(m); a.Ns = static_cast<int>(n_s); a.H = static_cast<int>(h);
  a.ldf = static_cast<int>(ldf); a.ldi = static_cast<int>(ldi); a.ldo = static_cast<int>(ldo); a.sigma = sigma;
#ifdef RDM_PC_TIMING
  a.clk = g_pc_clk;
#endif
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>(rdm_kpconv_fused_partial_rows(m, c));
  if (c == 1) {
    hipLaunchKernelGGL(kpconv_fused_c1_kernel, dim3(blocks), dim3(64 * kC1Waves), 0, st, a);
    return launch_status("kpconv_fused_c1_kernel");
  }
  if (!lockstep_form()) {  // one persistent workgroup per CU, producer / consumer wavefronts (> 64 KB of dynamic LDS)
    static std::atomic<uint64_t> pc32{0}, pc64{0};
    constexpr size_t l32 = fused_pc_lds_bytes<32, kPcNp32, kPcS32>(), l64 = fused_pc_lds_bytes<64, kPcNp64, kPcS64>();
    if (c == 32) {
      RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(kpconv_fused_pc_kernel<32, kPcNp32, kPcS32, kPcPf32, kPcWd32>),
                                        static_cast<int>(l32), pc32));
      hipLaunchKernelGGL((kpconv_fused_pc_kernel<32, kPcNp32, kPcS32, kPcPf32, kPcWd32>), dim3(blocks), dim3(1024), l32, st, a);
    } else {
      RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(kpconv_fused_pc_kernel<64, kPcNp64, kPcS64, kPcPf64, kPcWd64>),
                                        static_cast<int>(l64), pc64));
      hipLaunchKernelGGL((kpconv_fused_pc_kernel<64, kPcNp64, kPcS64, kPcPf64, kPcWd64>), dim3(blocks), dim3(1024), l64, st, a);
    }
    return launch_status("kpconv_fused_pc_kernel");
  }
  // (the C = 64 instance needs > 64 KB of dynamic LDS: the attribute is set once per device)
  static std::atomic<uint64_t> attr32{0}, attr64{0};
  RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(kpconv_fused_kernel<32, kQb32, kNw32, kIters32, kPf32>),
                                    static_cast<int>(fused_lds_bytes<32, kQb32, kNw32>()), attr32));
  RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(kpconv_fused_kernel<64, kQb64, kNw64, kIters64, kPf64>),
                                    static_cast<int>(fused_lds_bytes<64, kQb64, kNw64>()), attr64));
  const size_t lds32 = fused_lds_bytes<32, kQb32, kNw32>(), lds64 = fused_lds_bytes<64, kQb64, kNw64>();
  if (c == 32)
    hipLaunchKernelGGL((kpconv_fused_kernel<32, kQb32, kNw32, kIters32, kPf32>), dim3(blocks), dim3(64 * kNw32), lds32, st, a);
  else
    hipLaunchKernelGGL((kpconv_fused_kernel<64, kQb64, kNw64, kIters64, kPf64>), dim3(blocks), dim3(64 * kNw64), lds64, st, a);
  return launch_status("kpconv_fused_kernel");
}

extern "C" size_t rdm_kpconv_fused_workspace_bytes(int64_t m, int64_t c_in, int64_t c_out) {
  const size_t nblk = static_cast<size_t>(rdm::ceil_div<int64_t>(m > 0 ? m : 1, 16));  // (an upper bound of the partial rows)
  return rdm::align_up(nblk * 2 * c_out * sizeof(double)) + rdm_group_norm_workspace_bytes(m, c_out) + 256;
}

// KPConv + the GroupNorm (+ activation) that follows it in every block of the backbone (modules.py:141-145, 205-207):
// conv_out receives the convolution, y = act(GroupNorm(conv_out)).  The statistics come from the fused kernel's epilogue.
extern "C" int rdm_kpconv_fused_group_norm(const float* q_points, int64_t m, const float* s_points, int64_t n_s,
                                           const float* s_feats, int64_t c, int64_t ldf, const uint8_t* s_positive,
                                           const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                                           const float* kernel_points, float sigma, const float* w_packed, const float* bias,
                                           int64_t c_out, int groups, const float* gamma, const float* beta, float eps, int act,
                                           float* conv_out, int64_t ld_conv, float* y, int64_t ldy, void* ws, size_t ws_bytes,
                                           void* stream) {
  using namespace rdm;
  RDM_REQUIRE(gamma && beta && conv_out && y, "rdm_kpconv_fused_group_norm: null pointer");
  if (m == 0) return RDM_OK;
  Arena ar(ws, ws_bytes);
  const int nblk = static_cast<int>(rdm_kpconv_fused_partial_rows(m, c));
  double* partial = ar.take<double>(static_cast<size_t>(nblk) * 2 * c_out);
  const size_t gn_ws = rdm_group_norm_workspace_bytes(m, c_out);
  char* nws = ar.take<char>(gn_ws);
  if (!ar.ok) {
    set_error("rdm_kpconv_fused_group_norm: workspace too small (%zu < %zu)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  if (int e = rdm_kpconv_fused(q_points, m, s_points, n_s, s_feats, c, ldf, s_positive, idx, h, ldi, width, kernel_points, sigma,
                               w_packed, bias, c_out, conv_out, ld_conv, partial, stream))
    return e;
  return group_norm_finish(partial, nblk, conv_out, m, c_out, ld_conv, groups, gamma, beta, eps, nullptr, 0, act, y, ldy, nullptr,
                           nws, gn_ws, stream);
}
